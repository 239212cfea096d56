/*
 * vfsms.h -- C ABI of libvfsms.so, the MI355X (gfx950) engine for the VFSMS pairwise-alignment hot path.
 *
 * Plain C: opaque context pointer, plain pointers and sizes, int status codes.  No torch / numpy / C++
 * types cross this boundary.  Every entry point names the reference interface it replaces
 * (paths relative to the reference repository Keep-Passion/ImageStitch).
 *
 * The reference's own FFI for this path is the Boost.Python module `myGpuFeatures`
 * (appendix/myGpuFeatures.cpp:203-209: detectAndDescribeBySurf, detectAndDescribeByOrb, matchDescriptors),
 * selected by Method.isGPUAvailable (ImageUtility.py:254,265,285,304); the remaining arithmetic of the path
 * is reached through cv2 (SURF_create/detectAndCompute ImageUtility.py:258,262; DescriptorMatcher
 * ImageUtility.py:288-299; cv2.phaseCorrelate Stitcher.py:230) and numpy (ImageFusion.py:43-244).
 *
 * Conventions
 *   - return value: 0 = VFSMS_OK, negative = error (vfsms_last_error gives the message, thread-local).
 *   - the caller owns every host buffer and pre-allocates outputs with a capacity; the library never
 *     returns memory it owns and never frees caller memory.
 *   - all entry points are synchronous at return (host outputs are valid) unless stated otherwise;
 *     work is issued on the context's own HIP stream.
 *   - one context per thread/GPU; contexts share no mutable state.
 *   - "zero keypoints / zero matches" is NOT an error (n_out = 0).
 */
#ifndef VFSMS_H
#define VFSMS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFSMS_OK 0
#define VFSMS_ERR_BAD_ARG (-1)
#define VFSMS_ERR_CAPACITY (-2)   /* an output or internal capacity was exceeded            */
#define VFSMS_ERR_HIP (-3)        /* a HIP runtime call failed (message has hipGetErrorString) */
#define VFSMS_ERR_FFT (-4)        /* rocFFT plan / execution failure                           */
#define VFSMS_ERR_NO_DEVICE (-5)
#define VFSMS_ERR_UNSUPPORTED (-6)

typedef struct vfsms_ctx vfsms_ctx;

/* cv::KeyPoint fields SURF/ORB fill (the reference keeps only pt: ImageUtility.py:264) */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} vfsms_keypoint;

/* SURF parameters: cv2.xfeatures2d.SURF_create() defaults at ImageUtility.py:258 are
 * {100, 4, 3, extended 0, upright 0}; the DLL path passes Method.surf* (ImageUtility.py:23-28,272). */
typedef struct {
    float hessian_threshold;
    int32_t n_octaves;
    int32_t n_octave_layers;
    int32_t extended;     /* 0 -> 64-d, 1 -> 128-d */
    int32_t upright;
} vfsms_surf_params;

/* ORB parameters: cv2.ORB_create(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, WTA_K, scoreType, patchSize,
 * fastThreshold) as built at ImageUtility.py:260 from Method.orb* (ImageUtility.py:30-39): {5000, 1.2, 8, 31, 0, 2, 0, 31, 20}.
 * Supported: first_level 0, wta_k 2, score_type 0 (HARRIS), n_levels <= 8, patch_size <= 31.                                */
typedef struct {
    int32_t n_features;
    float scale_factor;
    int32_t n_levels, edge_threshold, first_level, wta_k, score_type, patch_size, fast_threshold;
} vfsms_orb_params;

/* One ROI attempt of the incremental search (Stitcher.py:319-351): ROI rectangles inside two
 * device-resident tiles, as Method.getROIRegionForIncreMethod (ImageUtility.py:66-101) slices them. */
typedef struct {
    int64_t tile_a, tile_b;          /* handles from vfsms_tile_upload / vfsms_tile_wrap */
    int32_t ay0, ax0, by0, bx0;      /* top-left corner of each ROI inside its tile      */
    int32_t h, w;                    /* common ROI size                                   */
} vfsms_roi_pair;

/* result of one feature attempt: out[8] = {status, dx, dy, votes, nA, nB, nMatches, reserved}
 * (status, [dx,dy]) is exactly Method.getOffsetByMode's return (ImageUtility.py:139-178)
 * BEFORE the stitch-axis correction of Stitcher.py:352-360 (that stays on the host).          */
#define VFSMS_ATTEMPT_INTS 8

/* ---- library / context -------------------------------------------------------------------------- */
int vfsms_version(void);
int vfsms_device_count(void);
int vfsms_last_error(char *buf, int buflen);           /* copies the calling thread's last message */
int vfsms_ctx_create(int device, vfsms_ctx **out);
int vfsms_ctx_destroy(vfsms_ctx *ctx);
int vfsms_ctx_sync(vfsms_ctx *ctx);
/* wait for the asynchronous tile uploads only (vfsms_tile_upload_async): their host buffers may be reused afterwards             */
int vfsms_ctx_sync_uploads(vfsms_ctx *ctx);
/* The context's hipStream_t (as void*), so a host framework can record HIP events on it.          */
void *vfsms_ctx_stream(vfsms_ctx *ctx);
/* Max SURF candidates per ROI (default: h*w/24 + 4096).  0 restores the default.                  */
int vfsms_ctx_set_keypoint_capacity(vfsms_ctx *ctx, int cap);

/* Per-stage timing with HIP events recorded on the context's own stream around each kernel group
 * ("integral", "hessian", "nms", "sort", "orientation", "describe", "bf_l2", "vote", "phase", "fuse", ...).
 * read: comma-separated stage names, accumulated milliseconds and launch-group counts; reset != 0 clears. */
int vfsms_profile_enable(vfsms_ctx *ctx, int on);
int vfsms_profile_read(vfsms_ctx *ctx, char *names, int names_len, double *ms, int64_t *calls, int cap,
                       int *n_out, int reset);

/* ---- device-resident tiles (grayscale u8, row stride in bytes) ----------------------------------- */
int vfsms_tile_upload(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride, int64_t *handle);
/* the same without waiting for the copy: it runs on the context's copy stream, overlapped with whatever the compute stream is doing,
 * and the first batch / canvas call that names the tile orders itself behind it (the decode loop of Stitcher.py:68-69 feeding the
 * GPU while earlier pairs are being registered).  img must stay valid and unchanged until vfsms_ctx_sync or the first synchronous
 * call that used the tile has returned; copies from pinned memory (vfsms_host_alloc) run at PCIe rate and truly overlap.          */
int vfsms_tile_upload_async(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride, int64_t *handle);
/* an interleaved colour tile (ch = 3: the isColorMode = True default of Main.py, Stitcher.py:174-179) for the mosaic canvas only:
 * rows of w * ch bytes, stride_bytes between rows, synchronous or (async != 0) on the copy stream like vfsms_tile_upload_async.
 * Registration entry points reject tiles with ch != 1.                                                                          */
int vfsms_tile_upload_ch(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int ch, int stride_bytes, int async, int64_t *handle);
/* Ingest pipeline (replaces the decode loop of Stitcher.py:68-69, which reads every file before the first pair is registered):
 * vfsms_tile_reserve hands out the handle of an h x w gray tile whose pixels arrive later; vfsms_tile_fill -- callable from ANY
 * thread, concurrently with a batch call on the context's thread -- copies them (it returns when the copy has landed, so the
 * decoder's staging buffer is free again; img == NULL reports a failed decode).  A batch / canvas call that names a reserved tile
 * waits for exactly that tile, so vfsms_pairs_offsets works on tiles 0, 1, ... while tile k is still being decoded.               */
int vfsms_tile_reserve(vfsms_ctx *ctx, int h, int w, int64_t *handle);
int vfsms_tile_fill(vfsms_ctx *ctx, int64_t handle, const uint8_t *img, int stride);
/* the same for an interleaved tile of ch channels (the mosaic's colour tiles); vfsms_tile_fill takes its rows of w * ch bytes, `stride` in bytes */
int vfsms_tile_reserve_ch(vfsms_ctx *ctx, int h, int w, int ch, int64_t *handle);
/* ONE decode per file for both uses the reference makes of it: cv2.imdecode(..., 0) feeds the registration loop (Stitcher.py:68-69) and,
 * with isColorMode (Main.py:14's default), cv2.imdecode(..., IMREAD_COLOR) feeds the mosaic (Stitcher.py:382-403).  Both are views of the
 * same entropy decode: IMREAD_GRAYSCALE of a JPEG is its Y plane, IMREAD_COLOR is libjpeg's fixed-point YCbCr -> RGB (jdcolor.c) of the
 * same planes, stored B G R.  `src` holds what the decoder produced once --
 *   VFSMS_SRC_GRAY8   one byte per pixel (a grayscale file; the colour tile replicates it, as IMREAD_COLOR does)
 *   VFSMS_SRC_YCC24   Y Cb Cr interleaved (libjpeg out_color_space = JCS_YCbCr: no colour conversion on the host)
 *   VFSMS_SRC_YCCX32  Y Cb Cr X, four bytes per pixel (Pillow's pixel storage, handed over without a repack)
 * -- and the device writes the reserved gray tile `gray` and the reserved 3-channel tile `color` (either may be 0).  Any thread; returns
 * when both tiles are complete; src == NULL gives both up (a failed decode).                                                              */
#define VFSMS_SRC_GRAY8 0
#define VFSMS_SRC_YCC24 1
#define VFSMS_SRC_YCCX32 2
int vfsms_tile_fill_pair(vfsms_ctx *ctx, int64_t gray, int64_t color, const uint8_t *src, int stride_bytes, int format);
/* The decode itself, for JPEG files (what cv2.imread / cv2.imdecode do on the reference's host: Stitcher.py:68-69, 382-403): the FILE'S
 * BYTES in, both tiles out.  The system's libjpeg-turbo (libjpeg.so.8, loaded at first use) decodes straight into the library's pinned
 * staging memory -- the Y plane alone when only `gray` is given (IMREAD_GRAYSCALE); when `color` is given, the DOWNSAMPLED Y Cb Cr planes of
 * a 4:2:0 file (round 6: libjpeg's h2v2 fancy upsampling and jdcolor.c's conversion both run on the device, k_ingest_420;
 * VFSMS_JPEG_RAW420=0 restores the host's upsampling), the upsampled planes of any other sampling -- and the device finishes as in
 * vfsms_tile_fill_pair.  Any thread; returns when both tiles are complete.  VFSMS_ERR_UNSUPPORTED (no
 * libjpeg.so.8 on this host; not a 1- or 3-component YCbCr / gray JPEG) and VFSMS_ERR_BAD_ARG (a damaged file; a file that is not the
 * size of the reserved tiles) leave BOTH TILES RESERVED: decode some other way and fill them, or give them up.                          */
int vfsms_tile_fill_jpeg(vfsms_ctx *ctx, int64_t gray, int64_t color, const uint8_t *jpeg, size_t nbytes);
/* the same decode into the caller's memory, no context and no GPU: rows of *w * *comp bytes, comp = 3 (Y Cb Cr interleaved) when
 * want_planes != 0 and the file has three components, else 1 (the grayscale decode).  VFSMS_ERR_CAPACITY (with *h, *w, *comp set) when
 * `cap` bytes are too few -- call with out == NULL to ask for the size.  want_planes == 2 (round 6): the DOWNSAMPLED planes of a 4:2:0
 * Y Cb Cr file as jpeg_read_raw_data hands them out (*comp = 420): Y with pitch pw = *w rounded up to 16 and ph = *h rounded up to 16 rows,
 * then Cb and Cr (pitch pw / 2, ph / 2 rows) -- pw * ph * 3 / 2 bytes; what vfsms_tile_fill_jpeg stages when the device does the chroma
 * upsampling (VFSMS_ERR_UNSUPPORTED for any other sampling).                                                                            */
int vfsms_jpeg_decode(const uint8_t *jpeg, size_t nbytes, int want_planes, uint8_t *out, size_t cap, int *h, int *w, int *comp);
/* The way out, for the mosaic: replaces cv2.imwrite(path, result) for .jpg results (Stitcher.py:149, 175-179; Main.py:21-51 writes every
 * result as jpg).  vfsms_jpeg_encode: n_rows x cols pixels of 1 or 3 channels (R G B, or B G R -- the canvas order -- with bgr != 0), rows
 * `stride_bytes` apart -> a complete baseline JPEG stream with cv2.imwrite's settings (libjpeg defaults, `quality`: 95).  No context, no GPU,
 * any thread.  *nbytes = the stream's size, also on VFSMS_ERR_CAPACITY.  A horizontal STRIPE of an image encodes independently of the
 * others when its height is a multiple of the MCU height (16 rows for colour, 8 for gray), so a host encodes the stripes of a mosaic on all
 * its cores -- while the bands are still leaving the device -- and vfsms_jpeg_join makes ONE file of them: the headers of stripe 0 with the
 * full height, a DRI segment (restart interval = the MCUs of a stripe, <= 65535) and the stripes' entropy-coded segments separated by RSTn
 * markers.  Same DCT coefficients, hence the same decoded pixels, as the one-thread encode of the whole image.  out == NULL: size only.   */
int vfsms_jpeg_encode(const uint8_t *rows, int n_rows, int cols, int channels, int stride_bytes, int bgr, int quality,
                      uint8_t *out, size_t cap, size_t *nbytes);
int vfsms_jpeg_join(const uint8_t *const *streams, const size_t *sizes, int n_stripes, int stripe_rows, int total_rows,
                    uint8_t *out, size_t cap, size_t *nbytes);
/* pinned host staging memory for tiles (decoders write into it; uploads from it are asynchronous DMA)                          */
int vfsms_host_alloc(vfsms_ctx *ctx, size_t bytes, void **ptr);
int vfsms_host_free(vfsms_ctx *ctx, void *ptr);
/* adopt memory already on this device (e.g. a framework tensor); not freed by vfsms_tile_free     */
int vfsms_tile_wrap(vfsms_ctx *ctx, const void *device_ptr, int h, int w, int stride, int64_t *handle);
int vfsms_tile_free(vfsms_ctx *ctx, int64_t handle);

/* ---- per-operator entry points, host buffers in / out -------------------------------------------- */
/* cv::integral(CV_8U -> CV_32S) as SURF uses it; sum_out is (h+1) x (w+1) int32                   */
int vfsms_integral_u8_i32(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride, int32_t *sum_out);

/* replaces myGpuFeatures.detectAndDescribeBySurf (appendix/myGpuFeatures.cpp:67-104) and
 * cv2 SURF detectAndCompute (ImageUtility.py:258,262).  kps_xy: float32[cap][2] = (x, y);
 * desc: float32[cap][64|128]; kps_full optional (may be NULL).                                    */
int vfsms_surf_detect_describe(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride,
                               const vfsms_surf_params *params,
                               float *kps_xy, float *desc, vfsms_keypoint *kps_full,
                               int cap, int *n_out);
/* detector only (sorted keypoints before orientation/deletion); for staged parity checks          */
int vfsms_surf_detect(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride,
                      const vfsms_surf_params *params, vfsms_keypoint *kps_full, int cap, int *n_out);

/* replaces myGpuFeatures.detectAndDescribeByOrb (appendix/myGpuFeatures.cpp:106-146) and cv2 ORB detectAndCompute
 * (ImageUtility.py:260,262).  kps_xy: float32[cap][2]; desc: uint8[cap][32]; kps_full optional.                   */
int vfsms_orb_detect_describe(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride,
                              const vfsms_orb_params *params,
                              float *kps_xy, uint8_t *desc, vfsms_keypoint *kps_full, int cap, int *n_out);

/* replaces myGpuFeatures.matchDescriptors(featureType 1|2, param=ratio) (appendix/myGpuFeatures.cpp:160-173)
 * and BFMatcher("BruteForce").knnMatch(k=2) + ratio filter (ImageUtility.py:288-296).
 * pairs: int32[cap][2] = (trainIdx, queryIdx) in query order.
 * Results are those of the reference's float arithmetic (4-wide accumulation of (a-b)^2, sqrt-domain compares, ties to the
 * lower train index) for every input.  64-d inputs whose rows all have norm <= 1 (SURF descriptors are L2-normalised; checked
 * on the device) are searched with the split-bf16 MFMA candidate filter + exact verification, anything else with the exhaustive
 * kernel; VFSMS_BF_EXACT=1 in the environment forces the exhaustive kernel.                                                  */
int vfsms_bf_l2_knn2_ratio(vfsms_ctx *ctx, const float *q, int nq, const float *t, int nt, int dim,
                           double ratio, int32_t *pairs, int cap, int *m_out);
/* raw 2-NN (for parity checks): idx1/d1 best, d2 second-best distance (+inf if nt < 2)            */
int vfsms_bf_l2_knn2(vfsms_ctx *ctx, const float *q, int nq, const float *t, int nt, int dim,
                     int32_t *idx1, float *d1, float *d2);
/* replaces matchDescriptors(featureType 3, param=orbMaxDistance) (appendix/myGpuFeatures.cpp:175-187)
 * and BFMatcher("BruteForce-Hamming").match (ImageUtility.py:297-302).  max_dist < 0: no threshold. */
int vfsms_bf_hamming_nn(vfsms_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes,
                        int max_dist, int32_t *pairs, int cap, int *m_out);

/* Method.getOffsetByMode (ImageUtility.py:139-178).  kps: float32[n][2]=(x,y); out4={status,dx,dy,votes} */
int vfsms_mode_offset(vfsms_ctx *ctx, const float *kpsA, int nA, const float *kpsB, int nB,
                      const int32_t *pairs, int m, int offset_evaluate, int32_t *out4);

/* cv2.phaseCorrelate(np.float64(a), np.float64(b)) (Stitcher.py:230): out3 = {x, y, response}     */
int vfsms_phase_correlate_u8(vfsms_ctx *ctx, const uint8_t *a, const uint8_t *b, int h, int w,
                             int stride_a, int stride_b, double *out3);
/* How a strip of h x w is correlated (no reference counterpart: what bench.py / the tests report about the path that ran).
 * info8 = {1 = the transforms run in LDS (csrc/phase_kernels.hip) | 0 = rocFFT plans, 1 = correlated as its byte transpose,
 *          column length M, row length N (the padded sizes in the orientation used), columns per workgroup, rows per workgroup,
 *          threads of a row workgroup, threads of a column workgroup}; no device work                                           */
int vfsms_phase_plan(int h, int w, int32_t *info8);

/* ImageFusion.fuseByFadeInAndFadeOut([A,B],dx,dy) (ImageFusion.py:192-244) on the reference's own
 * representation: int64 [r][c][ch] with -1 = empty (Stitcher.py:434-436).  out: uint8 [r][c][ch].
 * info (optional, 4 ints) = {mode 0 strip / 1 corner, corner index, rowIndex, colIndex}.           */
int vfsms_fuse_fade_i64(vfsms_ctx *ctx, const int64_t *A, const int64_t *B, int r, int c, int ch,
                        int dx, int dy, uint8_t *out, int32_t *info);
/* ImageFusion.fuseByTrigonometric([A,B],dx,dy) (ImageFusion.py:246-293), same representation and geometry decisions, weights
 * sin(w pi / 2)^2 of the float64 strip ramps / of getWeightsMatrix's float32 matrix.  The bytes hang on numpy's own sin in the
 * last ulp (where A == B the truncated result sits on an integer): equal to the reference on > 99.9 % of the bytes, |diff| <= 1.  */
int vfsms_fuse_trig_i64(vfsms_ctx *ctx, const int64_t *A, const int64_t *B, int r, int c, int ch,
                        int dx, int dy, uint8_t *out, int32_t *info);

/* The separable float32 ramps behind that blend, without blending: ramps = [wA_r(r) | wB_r(r) | wA_c(c) | wB_c(c)].
 * force_corner != 0 -> ImageFusion.getWeightsMatrix (ImageFusion.py:43-190): weightMatB = wB_r x wB_c,
 * weightMatA = 1 - weightMatB.  Otherwise the mode fuseByFadeInAndFadeOut itself would pick (info[0]).   */
int vfsms_fuse_ramps_i64(vfsms_ctx *ctx, const int64_t *A, int r, int c, int ch, int dx, int dy,
                         int force_corner, float *ramps, int32_t *info);

/* ---- fused, device-resident fast path ------------------------------------------------------------- */
/* n independent SURF + BF-L2 + ratio + mode-vote attempts in one batch (one launch sequence for all):
 * the body of the while-loop at Stitcher.py:322-343 for n (pair, direction, i) candidates.
 * out: int32[n][VFSMS_ATTEMPT_INTS].                                                               */
int vfsms_attempt_surf_batch(vfsms_ctx *ctx, const vfsms_roi_pair *jobs, int n,
                             const vfsms_surf_params *params, double ratio, int offset_evaluate,
                             int32_t *out);
/* the same with the reference's pre-enhancement of every ROI strip (Method.isEnhance, Stitcher.py:327-334):
 * enhance_mode 1 = cv2.equalizeHist, 2 = cv2.createCLAHE(clip_limit, (tile_grid, tile_grid)).apply, 0 = none                  */
int vfsms_attempt_surf_batch_enhanced(vfsms_ctx *ctx, const vfsms_roi_pair *jobs, int n, const vfsms_surf_params *params,
                                      double ratio, int offset_evaluate, int enhance_mode, double clip_limit, int tile_grid,
                                      int32_t *out);
/* same with ORB + BF-Hamming 1-NN (max_dist < 0: no distance threshold, the cv2 path; else distance < max_dist, the DLL path) */
int vfsms_attempt_orb_batch(vfsms_ctx *ctx, const vfsms_roi_pair *jobs, int n,
                            const vfsms_orb_params *params, int max_dist, int offset_evaluate, int32_t *out);
/* same for phase correlation (Stitcher.py:224-235): out: double[n][3] = {x, y, response}           */
int vfsms_attempt_phase_batch(vfsms_ctx *ctx, const vfsms_roi_pair *jobs, int n, double *out);

/* ---- whole shooting paths behind one call (the pair loop of Stitcher.flowStitch, Stitcher.py:64-79, around the incremental search of
 * Stitcher.py:306-367 / 205-258) -------------------------------------------------------------------------------------------------
 * The candidate state machine runs inside the library: speculative fused batches (a window of consecutive pairs at the inherited
 * direction, whole candidate rings at predicted turns), results selected in the reference's order
 *     for i in 1 .. maxI-1 { d = direction; do { attempt(d, i); d = rotate(d) } while (d != direction) },  maxI = floor(0.5 / roiRatio) + 2,
 * `direction` threaded from pair to pair (Stitcher.py:252,361).  out: int32[last_pair - first_pair][6] = {status, dx, dy, accepted
 * direction, ROI growth i, votes}, offsets already carried back to full-tile coordinates (Stitcher.py:352-360); direction_out: the
 * direction the path leaves with; stats (optional, int64[8]): attempts evaluated, fused batches, keypoint-capacity retries, sum nA*nB,
 * sum nA+nB, ROI pixels processed (workload figures for roofline reports), 2 reserved.
 * midpath != 0: the chain starts inside a path (a rank of the pair-sharded form): speculation stays short until two runs were seen.
 * stop_on_fail != 0: stop behind the first pair that cannot be registered (flowStitch breaks there, Stitcher.py:74-76).              */
typedef struct { int32_t pair, direction, i; } vfsms_attempt_key;
typedef struct {
    int32_t method;                  /* 0 SURF + BF-L2 + ratio + mode, 1 ORB + BF-Hamming + mode, 2 FP64 phase correlation        */
    int32_t offset_evaluate;         /* Method.offsetEvaluate                                                                      */
    int32_t direct_incre;            /* Stitcher.directIncre: 1, 0 or -1                                                           */
    int32_t window;                  /* speculation window (attempts per fused batch)                                              */
    int32_t orb_max_dist;            /* < 0: no Hamming threshold (the cv2 path)                                                   */
    int32_t enhance_mode;            /* Method.isEnhance: 0 none, 1 equalizeHist, 2 CLAHE (SURF only)                              */
    int32_t tile_grid;               /* Method.tileSize                                                                            */
    int32_t reserved;
    double roi_ratio;                /* Method.roiRatio                                                                            */
    double search_ratio;             /* Method.searchRatio (ratio test)                                                            */
    double phase_threshold;          /* Stitcher.phaseResponseThreshold                                                            */
    double clip_limit;               /* Method.clipLimit                                                                           */
    vfsms_surf_params surf;
    vfsms_orb_params orb;
    /* optional PREDICTION of the accepted direction of every pair of the path (the stage's scan pattern: e.g. a column serpentine of known
     * height), path_hint[k] in 1..4 for pair k, or NULL.  It only primes the speculation predictor of a chain that starts inside the path
     * (a rank of the pair-sharded form knows nothing of the runs before its chunk) -- results never depend on it.                       */
    const int32_t *path_hint;
    int32_t path_hint_len;
    int32_t reserved2;
} vfsms_grid_params;
int vfsms_pairs_offsets(vfsms_ctx *ctx, const int64_t *tiles, const int32_t *shapes_hw, int n_tiles, int first_pair, int last_pair,
                        int direction_in, int midpath, int stop_on_fail, const vfsms_grid_params *p, int32_t *out,
                        int32_t *direction_out, int64_t *stats);
/* The same state machine over a caller-supplied evaluator of fused batches (other operators; the CPU tests): eval fills
 * rows[n][VFSMS_ATTEMPT_INTS] = {status, raw dx, raw dy, votes, nA, nB, ...} for n attempts and returns VFSMS_OK.                   */
typedef int (*vfsms_attempt_eval)(void *user, const vfsms_attempt_key *items, int n, int32_t *rows);
int vfsms_pairs_offsets_eval(vfsms_attempt_eval eval, void *user, const int32_t *shapes_hw, int n_tiles, int first_pair, int last_pair,
                             int direction_in, int midpath, int stop_on_fail, const vfsms_grid_params *p, int32_t *out,
                             int32_t *direction_out, int64_t *stats);
/* A chunk in the MIDDLE of a path (a rank > 0 of the pair-sharded multi-GPU form): its incoming `self.direction` (Stitcher.py:252,361)
 * is the result of pairs another GPU is still registering, so the chunk [first_pair, last_pair) is registered for each of the four
 * possible incoming directions -- first candidates of the first pair in ONE batch, shared attempt cache, chains merged as soon as
 * they agree (after one pair, in practice).  out: int32[4][per][6], chain of incoming direction d at (d - 1) * per * 6 (per >= chunk
 * length: the padded row count of the all-gather payload); direction_out[4] = the direction each chain ends in.              */
int vfsms_pairs_offsets_blind(vfsms_ctx *ctx, const int64_t *tiles, const int32_t *shapes_hw, int n_tiles, int first_pair, int last_pair,
                              int per, const vfsms_grid_params *p, int32_t *out, int32_t *direction_out, int64_t *stats);
int vfsms_pairs_offsets_blind_eval(vfsms_attempt_eval eval, void *user, const int32_t *shapes_hw, int n_tiles, int first_pair, int last_pair,
                                   int per, const vfsms_grid_params *p, int32_t *out, int32_t *direction_out, int64_t *stats);

/* ---- whole-tile feature search with the reference's feature cache (Stitcher.calculateOffsetForFeatureSearch, Stitcher.py:260-304) --
 * Stitcher.tempImageFeature (Stitcher.py:14-18,278-290) keeps tile B's keypoints + descriptors so that they become tile A's of the
 * next pair; here that payload stays in HBM under a handle: SURF (+ optional enhancement, Stitcher.py:269-276) of a rectangle of a
 * resident tile -> feature handle; two handles -> BF-L2 2-NN + ratio + mode vote, only out[8] = {status, dx, dy, votes, nA, nB,
 * nMatches, 0} returns to the host.  vfsms_features_download gives the arrays Method.detectAndDescribe would have returned.      */
int vfsms_features_surf(vfsms_ctx *ctx, int64_t tile, int y0, int x0, int h, int w, const vfsms_surf_params *params,
                        int enhance_mode, double clip_limit, int tile_grid, int64_t *feat, int *n_out);
int vfsms_features_match_offset(vfsms_ctx *ctx, int64_t feat_a, int64_t feat_b, double ratio, int offset_evaluate, int32_t *out);
int vfsms_features_download(vfsms_ctx *ctx, int64_t feat, float *kps_xy, float *desc, int cap, int *n_out, int *dim_out);
/* The same for MANY tiles at once (the line scans of Main.py:29-51: 4 of the 6 demo datasets run calculateOffsetForFeatureSearch over
 * consecutive files, Stitcher.py:260-304): whole tiles, up to 16 per fused launch sequence, one host synchronisation per chunk;
 * feats[k] / counts[k] receive the set of tiles[k].  vfsms_features_match_offset_batch matches n (query set, train set) jobs --
 * the N - 1 consecutive pairs of a scan -- and votes in ONE batch: out[8 k ..] as vfsms_features_match_offset.                 */
int vfsms_features_surf_batch(vfsms_ctx *ctx, const int64_t *tiles, int n, const vfsms_surf_params *params,
                              int enhance_mode, double clip_limit, int tile_grid, int64_t *feats, int *counts);
int vfsms_features_match_offset_batch(vfsms_ctx *ctx, const int64_t *feat_a, const int64_t *feat_b, int n, double ratio,
                                      int offset_evaluate, int32_t *out);
int vfsms_features_free(vfsms_ctx *ctx, int64_t feat);
/* cv2.equalizeHist(img) (mode 1) / cv2.createCLAHE(clip_limit, (tile_grid, tile_grid)).apply(img) (mode 2), Stitcher.py:269-276;
 * out: uint8 [h][w] contiguous                                                                                                     */
int vfsms_enhance_u8(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride, int mode, double clip_limit, int tile_grid,
                     uint8_t *out);

/* ---- device-resident mosaic canvas (Stitcher.getStitchByOffset, Stitcher.py:369-486) -------------- */
/* u8 canvas + validity plane instead of the reference's int64 / -1 sentinel                         */
int vfsms_canvas_create(vfsms_ctx *ctx, int rows, int cols, int ch, int64_t *handle);
/* the buffers of the canvas freed last stay with the context for the next canvas of the same size (released by the next free of another
 * canvas or with the context): a session's mosaics are of one size, and allocating 2 x rows x cols bytes costs more than assembling them      */
int vfsms_canvas_free(vfsms_ctx *ctx, int64_t handle);
/* plain paste of a host tile (u8 [h][w][ch]) at (y0, x0): Stitcher.py:444-451 ("notFuse" / tile 0)  */
int vfsms_canvas_paste(vfsms_ctx *ctx, int64_t canvas, const uint8_t *tile, int h, int w, int y0, int x0);
/* paste + fade-fuse of the ROI [ry0,ry1) x [rx0,rx1) (canvas coords) exactly as Stitcher.py:457-483 +
 * ImageFusion.py:192-244 do: A = canvas before paste, B = canvas after paste.  info as above.        */
int vfsms_canvas_fuse_tile(vfsms_ctx *ctx, int64_t canvas, const uint8_t *tile, int h, int w,
                           int y0, int x0, int ry0, int rx0, int ry1, int rx1,
                           int dx, int dy, int32_t *info);
/* the same with the blend chosen by `method`: 0 fadeInAndFadeOut, 1 trigonometric (Stitcher.fuseImage, Stitcher.py:488-525)  */
int vfsms_canvas_fuse_tile_m(vfsms_ctx *ctx, int64_t canvas, const uint8_t *tile, int h, int w,
                             int y0, int x0, int ry0, int rx0, int ry1, int rx1,
                             int dx, int dy, int method, int32_t *info);
/* paste + element-wise blend of the ROI for fuseMethod "average" / "maximum" / "minimum" (mode 0 / 1 / 2): ImageFusion.py:12-41
 * behind the zero / empty filling of Stitcher.fuseImage (Stitcher.py:498-504).                                              */
int vfsms_canvas_blend_tile(vfsms_ctx *ctx, int64_t canvas, const uint8_t *tile, int h, int w,
                            int y0, int x0, int ry0, int rx0, int ry1, int rx1, int mode);
/* the same two operations for a tile that is already resident in HBM (a handle from vfsms_tile_upload / vfsms_tile_wrap with
 * stride == w, e.g. the tiles the registration phase uploaded, or a colour tile from vfsms_tile_upload_ch): no host copy; the
 * tile's channel count must be the canvas's.
 * With info == NULL the fuse only enqueues work (no host synchronisation per tile); a degenerate corner geometry -- where the
 * reference's getWeightsMatrix raises -- is then latched in the canvas and reported by vfsms_canvas_download.               */
int vfsms_canvas_paste_tile(vfsms_ctx *ctx, int64_t canvas, int64_t tile, int y0, int x0);
/* vfsms_canvas_blend_tile ("average" / "maximum" / "minimum", mode 0 / 1 / 2) for a resident tile; enqueue only                       */
int vfsms_canvas_blend_tile_resident(vfsms_ctx *ctx, int64_t canvas, int64_t tile,
                                     int y0, int x0, int ry0, int rx0, int ry1, int rx1, int mode);
int vfsms_canvas_fuse_tile_resident(vfsms_ctx *ctx, int64_t canvas, int64_t tile,
                                    int y0, int x0, int ry0, int rx0, int ry1, int rx1,
                                    int dx, int dy, int32_t *info);
int vfsms_canvas_fuse_tile_resident_m(vfsms_ctx *ctx, int64_t canvas, int64_t tile,
                                      int y0, int x0, int ry0, int rx0, int ry1, int rx1,
                                      int dx, int dy, int method, int32_t *info);
/* The canvas walk of Stitcher.getStitchByOffset (Stitcher.py:434-483) over n resident tiles in one call.
 * geom: n x 9 ints [y0, x0, ry0, rx0, ry1, rx1, dx, dy, mode], mode -1 = paste (first tile / notFuse),
 * 0 = fadeInAndFadeOut, 1 = trigonometric, 2 / 3 / 4 = average / maximum / minimum.  Enqueue only: errors of a tile's geometry surface in the download. */
int vfsms_canvas_assemble_resident(vfsms_ctx *ctx, int64_t canvas, int n, const int64_t *tiles, const int32_t *geom);
/* final image: empty -> 0 (Stitcher.py:485-486).  out: u8 [rows][cols][ch]                           */
int vfsms_canvas_download(vfsms_ctx *ctx, int64_t canvas, uint8_t *out);
/* rows [row0, row0 + nrows) of the same image: a multi-GB mosaic leaves the device band by band and can be handed to an
 * incremental writer (the reference holds the whole int64 canvas and the u8 copy in host memory, Stitcher.py:434-436, 485-486) */
int vfsms_canvas_download_rows(vfsms_ctx *ctx, int64_t canvas, int row0, int nrows, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif /* VFSMS_H */

/*
 * vfsms_oracle.h -- CPU ORACLE for the VFSMS pairwise-alignment hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (imagestitch_amd/ +
 * libvfsms.so) never links, imports or calls anything in oracle/.
 *
 * It restates, in plain C, the arithmetic the reference (Keep-Passion/ImageStitch) reaches
 * through opencv-python==3.3.1.11 / opencv-contrib-python==3.3.1.11 (requirements.txt:113-114;
 * the OpenCV sources are NOT under /root/reference and cv2 is not installable here), plus the
 * reference's own numpy fuse (ImageFusion.py:43-244) and mode vote (ImageUtility.py:139-178).
 *
 * PARITY STATUS
 *   - fuse / getWeightsMatrix / getOffsetByMode : pinned against golden vectors captured by
 *     importing the reference's Python here (tools/capture_golden.py -> tests/golden/).
 *   - SURF / BFMatcher / phaseCorrelate        : "parity unpinned" against OpenCV itself (no cv2
 *     anywhere); pinned instead on (i) the 89-offset list at Stitcher.py:87 (+-1 px) on real
 *     dendriticCrystal strips, (ii) synthetic grids with exact integer ground truth, and
 *     (iii) OpenCV's published constants (layer sizes, optimal DFT sizes).
 */
#ifndef VFSMS_ORACLE_H
#define VFSMS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint as produced by SURF (pt.x, pt.y, size, angle, response, octave, class_id) */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

/* sin / cos in double with an explicit operation order (Cody-Waite reduction by pi/2 with a 33-bit head, fdlibm minimax
 * polynomials on [-pi/4, pi/4], Horner form).  The reference reaches std::sin/std::cos(float) for the SURF descriptor
 * window (upstream surf.cpp SURFInvoker; ImageUtility.py:262) and cos/sin(double) for ORB's rBRIEF rotation (upstream orb.cpp;
 * ImageUtility.py:260); a correctly rounded result is library independent, glibc's and the GPU's OCML routines are not.
 * The engine evaluates the same algorithm (imagestitch_amd/csrc/detmath.h, written from the same description), so both
 * sides round to the same float.  |error| < 1e-16: the float rounding equals correctly rounded sinf/cosf except within
 * ~2^-29 ulp of a rounding boundary.  Valid for the float-valued |x| <= 2^10 of this path. */
static inline void det_sincos(double x, double *s_out, double *c_out)
{
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;
    const double PIO2_LO = 6.07710050650619224932e-11;
    const double kd = __builtin_rint(x * INV_PIO2);
    const int k = (int)kd;
    const double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double sr = r + (z * r) * (S1 + z * ps);
    const double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double cr = 1.0 - (0.5 * z - z * pc);
    double s, c;
    switch (k & 3) {
    case 0: s = sr; c = cr; break;
    case 1: s = cr; c = -sr; break;
    case 2: s = -sr; c = -cr; break;
    default: s = -cr; c = sr; break;
    }
    *s_out = s; *c_out = c;
}
/* cv2.equalizeHist(img) and cv2.createCLAHE(clipLimit, (tilesX, tilesY)).apply(img) on 8-bit images (Stitcher.py:269-276,327-334);
 * dst is contiguous h x w.  OpenCV 3.3.1 histogram.cpp / clahe.cpp restated; parity unpinned against cv2. */
void orc_equalize_hist(const uint8_t *src, int h, int w, int stride, uint8_t *dst);
void orc_clahe(const uint8_t *src, int h, int w, int stride, double clipLimit, int tilesX, int tilesY, uint8_t *dst);

/* exported for tests: out[2n] = sin(x[n]), out[2n+1] = cos(x[n]) */
void orc_det_sincos(const double *x, int n, double *out);

/* cv::getOptimalDFTSize: smallest 2^a 3^b 5^c >= n  (Stitcher.py:230 via cv2.phaseCorrelate) */
int orc_optimal_dft_size(int n);

/* cv::integral(img, sum, CV_32S): sum is (h+1) x (w+1), row 0 / col 0 zero. */
void orc_integral_u8_i32(const uint8_t *img, int h, int w, int stride, int32_t *sum);

/* Fast-Hessian layer det/trace for one (size, step) layer: arrays are (h/step) x (w/step),
 * zero-initialised here (OpenCV leaves unwritten cells uninitialised; they are never read). */
void orc_surf_layer(const int32_t *sum, int h, int w, int size, int step, float *det, float *trace);

/* cv2.xfeatures2d.SURF_create(hess, nOct, nLayers, extended, upright).detectAndCompute(img, None)
 * (ImageUtility.py:258,262).  kps/desc are caller-allocated with capacity cap; desc is cap x (64|128).
 * Returns number of keypoints (after deletion of size<=0 ones), or -1 if cap was too small. */
int orc_surf_detect_describe(const uint8_t *img, int h, int w, int stride,
                             double hessianThreshold, int nOctaves, int nOctaveLayers,
                             int extended, int upright,
                             orc_keypoint *kps, float *desc, int cap, int nthreads);

/* Detector only (keypoints sorted by KeypointGreater, before orientation): for staged parity tests. */
int orc_surf_detect(const uint8_t *img, int h, int w, int stride,
                    double hessianThreshold, int nOctaves, int nOctaveLayers,
                    orc_keypoint *kps, int cap);

/* BFMatcher(NORM_L2).knnMatch(q, t, 2): per query the best train index, its distance and the
 * second-best distance (sqrt of float-accumulated squared L2; ties keep the lower train index).
 * For nt < 2, d2 is +inf and idx2 -1 (reference guards with len(m)==2, ImageUtility.py:294). */
void orc_bf_l2_knn2(const float *q, int nq, const float *t, int nt, int dim,
                    int32_t *idx1, float *d1, int32_t *idx2, float *d2, int nthreads);

/* ImageUtility.py:288-296: ratio filter in Python double arithmetic on the float32 distances.
 * pairs[m] = (trainIdx, queryIdx) in query order.  Returns M. */
int orc_bf_l2_ratio_matches(const float *q, int nq, const float *t, int nt, int dim,
                            double ratio, int32_t *pairs, int nthreads);

/* BFMatcher(NORM_HAMMING).match(q, t): one match per query, first minimum wins.
 * max_dist < 0: no threshold (CPU path, ImageUtility.py:297-302); else keep distance < max_dist
 * (GPU DLL path, myGpuFeatures.cpp:178-186).  Returns M. */
int orc_bf_hamming_matches(const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes,
                           int max_dist, int32_t *pairs, int32_t *dist_out);

/* Method.getOffsetByMode (ImageUtility.py:139-178).  kps are float32 [n][2] = (x, y).
 * out = {status, dx, dy, votes}. */
void orc_mode_offset(const float *kpsA, const float *kpsB, const int32_t *pairs, int m,
                     int offsetEvaluate, int32_t *out4);

/* cv2.phaseCorrelate(np.float64(a), np.float64(b)) as called at Stitcher.py:230.
 * out = {x, y, response}.  (M, N) padded size returned through MN if non-NULL. */
void orc_phase_correlate_u8(const uint8_t *a, const uint8_t *b, int h, int w,
                            int strideA, int strideB, double *out3, int *MN);

/* ImageFusion.fuseByFadeInAndFadeOut (ImageFusion.py:192-244) incl. getWeightsMatrix (:43-190).
 * A, B are int64 [r][c][ch] with -1 = empty (Stitcher.py:434-436).  A is modified in place exactly
 * as the reference does (imageA[imageA<0] = imageB[imageA<0]).  out is uint8 [r][c][ch].
 * info (optional, 4 ints) = {mode(0 strip,1 corner), corner index, rowIndex, colIndex}. */
void orc_fuse_fade(int64_t *A, const int64_t *B, int r, int c, int ch, int dx, int dy,
                   uint8_t *out, int32_t *info);

/* ImageFusion.getWeightsMatrix (ImageFusion.py:43-190): weightMatB = wB_r (x) wB_c, weightMatA = 1 - weightMatB. */
int orc_corner_ramps(const int64_t *A, int r, int c, int ch, float *wB_r, float *wB_c, int32_t *info);

/* cv2.ORB_create(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, 2, HARRIS_SCORE, patchSize, fastThreshold)
 * .detectAndCompute(image, None)  (ImageUtility.py:260,262; vfsms_oracle_orb.c states two deviations from upstream).
 * kps: x, y (level-0 coordinates), size = patchSize*scale, angle, response (Harris), octave; desc: uint8 [cap][32]. */
int orc_orb_detect_describe(const uint8_t *image, int h, int w, int stride,
                            int nfeatures, float scaleFactor, int nlevels, int edgeThreshold, int firstLevel,
                            int patchSize, int fastThreshold, orc_keypoint *kps, uint8_t *desc, int cap);
/* orb.cpp makeRandomPattern(patchSize, pattern, npoints): xy = int32[npoints][2] */
void orc_orb_pattern(int patchSize, int npoints, int32_t *xy);

#ifdef __cplusplus
}
#endif
#endif

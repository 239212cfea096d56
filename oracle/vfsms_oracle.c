/*
 * vfsms_oracle.c -- CPU ORACLE (test infrastructure only; see vfsms_oracle.h header note).
 *
 * Plain-C restatement of the arithmetic on the VFSMS hot path.  Every function cites the
 * reference call site it stands in for (paths relative to /root/reference) and, where the
 * arithmetic lives in the un-vendored dependency opencv(-contrib)-python==3.3.1.11
 * (requirements.txt:113-114), the upstream file whose published algorithm is restated.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: the restated float/double operation order is the specification.
 */
#include "vfsms_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * small OpenCV helpers (core/fast_math.hpp): cvRound = round-half-even, cvFloor, cvCeil
 * ---------------------------------------------------------------------------------------- */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil_d(double v) { int i = (int)v; return i + (i < v); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* core/src/dxt.cpp getOptimalDFTSize: smallest 2^a*3^b*5^c >= n */
int orc_optimal_dft_size(int n)
{
    if (n <= 1) return 1;
    for (int m = n;; m++) {
        int k = m;
        while (k % 2 == 0) k /= 2;
        while (k % 3 == 0) k /= 3;
        while (k % 5 == 0) k /= 5;
        if (k == 1) return m;
    }
}

/* imgproc/src/sumpixels.cpp integral(CV_8U -> CV_32S); called from SURF_Impl::detectAndCompute,
 * reached from ImageUtility.py:262 */
void orc_integral_u8_i32(const uint8_t *img, int h, int w, int stride, int32_t *sum)
{
    int sw = w + 1;
    memset(sum, 0, sizeof(int32_t) * (size_t)sw);
    for (int y = 0; y < h; y++) {
        int32_t s = 0;
        int32_t *row = sum + (size_t)(y + 1) * sw;
        const int32_t *prev = sum + (size_t)y * sw;
        row[0] = 0;
        for (int x = 0; x < w; x++) {
            s += img[(size_t)y * stride + x];
            row[x + 1] = prev[x + 1] + s;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * SURF  (opencv_contrib 3.3.1 modules/xfeatures2d/src/surf.cpp), defaults of
 * cv2.xfeatures2d.SURF_create() at ImageUtility.py:258: hessian 100, 4 octaves, 3 layers,
 * extended=False (64-d), upright=False.
 * ---------------------------------------------------------------------------------------- */
#define SURF_HAAR_SIZE0 9
#define SURF_HAAR_SIZE_INC 6
#define ORI_RADIUS 6
#define ORI_WIN 60
#define PATCH_SZ 20
#define SURF_ORI_SEARCH_INC 5
#define SURF_ORI_SIGMA 2.5f
#define SURF_DESC_SIGMA 3.3f

typedef struct { int p0, p1, p2, p3; float w; } surf_hf;

/* surf.cpp resizeHaarPattern */
static void resize_haar(const int src[][5], surf_hf *dst, int n, int oldSize, int newSize, int widthStep)
{
    float ratio = (float)newSize / oldSize;
    for (int k = 0; k < n; k++) {
        int dx1 = cv_round_f(ratio * src[k][0]);
        int dy1 = cv_round_f(ratio * src[k][1]);
        int dx2 = cv_round_f(ratio * src[k][2]);
        int dy2 = cv_round_f(ratio * src[k][3]);
        dst[k].p0 = dy1 * widthStep + dx1;
        dst[k].p1 = dy2 * widthStep + dx1;
        dst[k].p2 = dy1 * widthStep + dx2;
        dst[k].p3 = dy2 * widthStep + dx2;
        dst[k].w = src[k][4] / ((float)(dx2 - dx1) * (dy2 - dy1));
    }
}

/* surf.cpp calcHaarPattern: int box sum * float weight (float product), accumulated in double */
static inline float calc_haar(const int32_t *origin, const surf_hf *f, int n)
{
    double d = 0;
    for (int k = 0; k < n; k++)
        d += (origin[f[k].p0] + origin[f[k].p3] - origin[f[k].p1] - origin[f[k].p2]) * f[k].w;
    return (float)d;
}

/* surf.cpp calcLayerDetAndTrace.  sum is (h+1)x(w+1). det/trace are (h/step)x(w/step). */
void orc_surf_layer(const int32_t *sum, int h, int w, int size, int step, float *det, float *trace)
{
    static const int dx_s[3][5] = { {0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1} };
    static const int dy_s[3][5] = { {2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1} };
    static const int dxy_s[4][5] = { {1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1} };
    int sw = w + 1;
    int lrows = h / step, lcols = w / step;
    memset(det, 0, sizeof(float) * (size_t)lrows * lcols);
    memset(trace, 0, sizeof(float) * (size_t)lrows * lcols);
    if (size > h || size > w) return;
    surf_hf Dx[3], Dy[3], Dxy[4];
    resize_haar(dx_s, Dx, 3, 9, size, sw);
    resize_haar(dy_s, Dy, 3, 9, size, sw);
    resize_haar(dxy_s, Dxy, 4, 9, size, sw);
    int samples_i = 1 + (h - size) / step;
    int samples_j = 1 + (w - size) / step;
    int margin = (size / 2) / step;
    for (int i = 0; i < samples_i; i++) {
        const int32_t *sp = sum + (size_t)(i * step) * sw;
        float *dp = det + (size_t)(i + margin) * lcols + margin;
        float *tp = trace + (size_t)(i + margin) * lcols + margin;
        for (int j = 0; j < samples_j; j++) {
            float dx = calc_haar(sp, Dx, 3);
            float dy = calc_haar(sp, Dy, 3);
            float dxy = calc_haar(sp, Dxy, 4);
            sp += step;
            dp[j] = dx * dy - 0.81f * dxy * dxy;
            tp[j] = dx + dy;
        }
    }
}

/* surf.cpp interpolateKeypoint; Matx33f::solve(DECOMP_LU) = core/operations.hpp
 * Matx_FastSolveOp<float,3,3,1> (Cramer's rule in float, d = 1/det) */
static int interpolate_keypoint(float N9[3][9], int dx, int dy, int ds, orc_keypoint *kpt)
{
    float b0 = -(N9[1][5] - N9[1][3]) / 2;
    float b1 = -(N9[1][7] - N9[1][1]) / 2;
    float b2 = -(N9[2][4] - N9[0][4]) / 2;
    float a00 = N9[1][3] - 2 * N9[1][4] + N9[1][5];
    float a01 = (N9[1][8] - N9[1][6] - N9[1][2] + N9[1][0]) / 4;
    float a02 = (N9[2][5] - N9[2][3] - N9[0][5] + N9[0][3]) / 4;
    float a10 = a01;
    float a11 = N9[1][1] - 2 * N9[1][4] + N9[1][7];
    float a12 = (N9[2][7] - N9[2][1] - N9[0][7] + N9[0][1]) / 4;
    float a20 = a02;
    float a21 = a12;
    float a22 = N9[0][4] - 2 * N9[1][4] + N9[2][4];
    float x0 = 0, x1 = 0, x2 = 0;
    float det = a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
    float d = (float)(double)det;
    if (d != 0) {
        d = 1 / d;
        x0 = d * (b0 * (a11 * a22 - a12 * a21) - a01 * (b1 * a22 - a12 * b2) + a02 * (b1 * a21 - a11 * b2));
        x1 = d * (a00 * (b1 * a22 - a12 * b2) - b0 * (a10 * a22 - a12 * a20) + a02 * (a10 * b2 - b1 * a20));
        x2 = d * (a00 * (a11 * b2 - b1 * a21) - a01 * (a10 * b2 - b1 * a20) + b0 * (a10 * a21 - a11 * a20));
    }
    int ok = (x0 != 0 || x1 != 0 || x2 != 0) && fabsf(x0) <= 1 && fabsf(x1) <= 1 && fabsf(x2) <= 1;
    if (ok) {
        kpt->x += x0 * dx;
        kpt->y += x1 * dy;
        kpt->size = (float)cv_round_f(kpt->size + x2 * ds);
    }
    return ok;
}

typedef struct { orc_keypoint kp; int layer_index, i, j; } kp_cand;

/* surf.cpp KeypointGreater; final (layer,i,j) tie-break added only to make the order total */
static int kp_greater_cmp(const void *pa, const void *pb)
{
    const kp_cand *a = (const kp_cand *)pa, *b = (const kp_cand *)pb;
    if (a->kp.response > b->kp.response) return -1;
    if (a->kp.response < b->kp.response) return 1;
    if (a->kp.size > b->kp.size) return -1;
    if (a->kp.size < b->kp.size) return 1;
    if (a->kp.octave > b->kp.octave) return -1;
    if (a->kp.octave < b->kp.octave) return 1;
    if (a->kp.y < b->kp.y) return 1;       /* y DESCENDING: `if(kp1.pt.y < kp2.pt.y) return false; if(kp1.pt.y > kp2.pt.y) return true;` */
    if (a->kp.y > b->kp.y) return -1;
    if (a->kp.x < b->kp.x) return -1;
    if (a->kp.x > b->kp.x) return 1;
    if (a->layer_index != b->layer_index) return a->layer_index < b->layer_index ? -1 : 1;
    if (a->i != b->i) return a->i < b->i ? -1 : 1;
    if (a->j != b->j) return a->j < b->j ? -1 : 1;
    return 0;
}

/* surf.cpp fastHessianDetector + SURFFindInvoker::findMaximaInLayer (no mask) */
static int fast_hessian(const int32_t *sum, int h, int w, int nOctaves, int nOctaveLayers,
                        float hessianThreshold, kp_cand **out)
{
    int nTotal = (nOctaveLayers + 2) * nOctaves;
    float **dets = (float **)calloc(nTotal, sizeof(float *));
    float **traces = (float **)calloc(nTotal, sizeof(float *));
    int *sizes = (int *)calloc(nTotal, sizeof(int));
    int *steps = (int *)calloc(nTotal, sizeof(int));
    int index = 0, step = 1;
    for (int o = 0; o < nOctaves; o++) {
        for (int l = 0; l < nOctaveLayers + 2; l++) {
            size_t n = (size_t)(h / step) * (w / step);
            dets[index] = (float *)malloc(sizeof(float) * (n ? n : 1));
            traces[index] = (float *)malloc(sizeof(float) * (n ? n : 1));
            sizes[index] = (SURF_HAAR_SIZE0 + SURF_HAAR_SIZE_INC * l) << o;
            steps[index] = step;
            index++;
        }
        step *= 2;
    }
#pragma omp parallel for schedule(dynamic)
    for (int k = 0; k < nTotal; k++)
        orc_surf_layer(sum, h, w, sizes[k], steps[k], dets[k], traces[k]);

    int cap = 1024, n = 0;
    kp_cand *c = (kp_cand *)malloc(sizeof(kp_cand) * cap);
    for (int o = 0; o < nOctaves; o++) {
        for (int l = 1; l <= nOctaveLayers; l++) {
            int layer = o * (nOctaveLayers + 2) + l;
            int size = sizes[layer], ss = steps[layer];
            int lrows = h / ss, lcols = w / ss;
            int margin = (sizes[layer + 1] / 2) / ss + 1;
            int st = lcols;
            for (int i = margin; i < lrows - margin; i++) {
                const float *det_ptr = dets[layer] + (size_t)i * lcols;
                const float *trace_ptr = traces[layer] + (size_t)i * lcols;
                for (int j = margin; j < lcols - margin; j++) {
                    float val0 = det_ptr[j];
                    if (!(val0 > hessianThreshold)) continue;
                    int sum_i = ss * (i - (size / 2) / ss);
                    int sum_j = ss * (j - (size / 2) / ss);
                    const float *d1 = dets[layer - 1] + (size_t)i * lcols + j;
                    const float *d2 = dets[layer] + (size_t)i * lcols + j;
                    const float *d3 = dets[layer + 1] + (size_t)i * lcols + j;
                    float N9[3][9] = {
                        { d1[-st - 1], d1[-st], d1[-st + 1], d1[-1], d1[0], d1[1], d1[st - 1], d1[st], d1[st + 1] },
                        { d2[-st - 1], d2[-st], d2[-st + 1], d2[-1], d2[0], d2[1], d2[st - 1], d2[st], d2[st + 1] },
                        { d3[-st - 1], d3[-st], d3[-st + 1], d3[-1], d3[0], d3[1], d3[st - 1], d3[st], d3[st + 1] } };
                    int is_max = 1;
                    for (int a = 0; a < 3 && is_max; a++)
                        for (int b = 0; b < 9; b++) {
                            if (a == 1 && b == 4) continue;
                            if (!(val0 > N9[a][b])) { is_max = 0; break; }
                        }
                    if (!is_max) continue;
                    float center_i = sum_i + (size - 1) * 0.5f;
                    float center_j = sum_j + (size - 1) * 0.5f;
                    kp_cand kc;
                    kc.kp.x = center_j; kc.kp.y = center_i; kc.kp.size = (float)sizes[layer];
                    kc.kp.angle = -1; kc.kp.response = val0; kc.kp.octave = o;
                    kc.kp.class_id = (trace_ptr[j] > 0) - (trace_ptr[j] < 0);
                    kc.layer_index = layer; kc.i = i; kc.j = j;
                    int ds = size - sizes[layer - 1];
                    if (interpolate_keypoint(N9, ss, ss, ds, &kc.kp)) {
                        if (n == cap) { cap *= 2; c = (kp_cand *)realloc(c, sizeof(kp_cand) * cap); }
                        c[n++] = kc;
                    }
                }
            }
        }
    }
    qsort(c, n, sizeof(kp_cand), kp_greater_cmp);
    for (int k = 0; k < nTotal; k++) { free(dets[k]); free(traces[k]); }
    free(dets); free(traces); free(sizes); free(steps);
    *out = c;
    return n;
}

/* imgproc/src/smooth.cpp getGaussianKernel(n, sigma, CV_32F), sigma > 0 */
static void gaussian_kernel_f32(int n, double sigma, float *cf)
{
    double scale2X = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) cf[i] = (float)(cf[i] * sum);
}

/* core/src/mathfuncs_core atan_f32 / fastAtan2 (degrees) */
static inline float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* imgproc/src/resize.cpp  resize(win_size x win_size u8 -> 21x21 u8, INTER_AREA), shrink only
 * (win_size >= 25 always: SURF sizes >= 9 give s >= 1.2).  Three upstream paths:
 *   scale == 2 exactly   : ResizeAreaFastVec (a+b+c+d+2)>>2
 *   integer scale        : ResizeAreaFast_Invoker  int sum, saturate_cast<uchar>(sum * (1.f/area))
 *   otherwise            : computeResizeAreaTab + ResizeArea_Invoker (float accumulators) */
typedef struct { int si, di; float alpha; } dec_alpha;

static int area_tab(int ssize, int dsize, double scale, dec_alpha *tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale;
        double fsx2 = fsx1 + scale;
        double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = cv_ceil_d(fsx1), sx2 = cv_floor_d(fsx2);
        sx2 = imin(sx2, ssize - 1);
        sx1 = imin(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) {
            tab[k].di = dx; tab[k].si = sx1 - 1;
            tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth);
        }
        for (int sx = sx1; sx < sx2; sx++) {
            tab[k].di = dx; tab[k].si = sx;
            tab[k++].alpha = (float)(1.0 / cellWidth);
        }
        if (fsx2 - sx2 > 1e-3) {
            double m = fsx2 - sx2; if (m > 1.) m = 1.; if (m > cellWidth) m = cellWidth;
            tab[k].di = dx; tab[k].si = sx2;
            tab[k++].alpha = (float)(m / cellWidth);
        }
    }
    return k;
}

static inline uint8_t sat_u8_from_float(float v)
{
    int iv = cv_round_f(v);
    return (uint8_t)(iv < 0 ? 0 : iv > 255 ? 255 : iv);
}

static void resize_area_u8(const uint8_t *src, int ssz, uint8_t *dst /* 21x21 */)
{
    const int dsz = PATCH_SZ + 1;
    double inv_scale = (double)dsz / ssz;
    double scale = 1. / inv_scale;
    int iscale = cv_round_d(scale);
    int is_area_fast = fabs(scale - iscale) < DBL_EPSILON;
    if (is_area_fast) {
        if (iscale == 2) {
            for (int dy = 0; dy < dsz; dy++)
                for (int dx = 0; dx < dsz; dx++) {
                    const uint8_t *S = src + (size_t)(dy * 2) * ssz + dx * 2;
                    dst[dy * dsz + dx] = (uint8_t)((S[0] + S[1] + S[ssz] + S[ssz + 1] + 2) >> 2);
                }
            return;
        }
        int area = iscale * iscale;
        float fscale = 1.f / (area);
        for (int dy = 0; dy < dsz; dy++)
            for (int dx = 0; dx < dsz; dx++) {
                int sum = 0;
                for (int sy = 0; sy < iscale; sy++)
                    for (int sx = 0; sx < iscale; sx++)
                        sum += src[(size_t)(dy * iscale + sy) * ssz + dx * iscale + sx];
                dst[dy * dsz + dx] = sat_u8_from_float(sum * fscale);
            }
        return;
    }
    dec_alpha *tab = (dec_alpha *)malloc(sizeof(dec_alpha) * (size_t)ssz * 2);
    int tn = area_tab(ssz, dsz, scale, tab); /* same table for x and y (square) */
    float buf[PATCH_SZ + 1], sum[PATCH_SZ + 1];
    for (int dx = 0; dx < dsz; dx++) sum[dx] = 0;
    int prev_dy = tab[0].di;
    for (int j = 0; j < tn; j++) {
        float beta = tab[j].alpha;
        int dy = tab[j].di, sy = tab[j].si;
        const uint8_t *S = src + (size_t)sy * ssz;
        for (int dx = 0; dx < dsz; dx++) buf[dx] = 0;
        for (int k = 0; k < tn; k++) buf[tab[k].di] += S[tab[k].si] * tab[k].alpha;
        if (dy != prev_dy) {
            for (int dx = 0; dx < dsz; dx++) {
                dst[prev_dy * dsz + dx] = sat_u8_from_float(sum[dx]);
                sum[dx] = beta * buf[dx];
            }
            prev_dy = dy;
        } else {
            for (int dx = 0; dx < dsz; dx++) sum[dx] += beta * buf[dx];
        }
    }
    for (int dx = 0; dx < dsz; dx++) dst[prev_dy * dsz + dx] = sat_u8_from_float(sum[dx]);
    free(tab);
}

typedef struct {
    int nOriSamples;
    int aptx[169], apty[169];
    float aptw[169];
    float DW[PATCH_SZ * PATCH_SZ];
} surf_tables;

static void surf_tables_init(surf_tables *T)
{
    float G_ori[2 * ORI_RADIUS + 1], G_desc[PATCH_SZ];
    gaussian_kernel_f32(2 * ORI_RADIUS + 1, SURF_ORI_SIGMA, G_ori);
    T->nOriSamples = 0;
    for (int i = -ORI_RADIUS; i <= ORI_RADIUS; i++)
        for (int j = -ORI_RADIUS; j <= ORI_RADIUS; j++)
            if (i * i + j * j <= ORI_RADIUS * ORI_RADIUS) {
                T->aptx[T->nOriSamples] = i;
                T->apty[T->nOriSamples] = j;
                T->aptw[T->nOriSamples++] = G_ori[i + ORI_RADIUS] * G_ori[j + ORI_RADIUS];
            }
    gaussian_kernel_f32(PATCH_SZ, SURF_DESC_SIGMA, G_desc);
    for (int i = 0; i < PATCH_SZ; i++)
        for (int j = 0; j < PATCH_SZ; j++)
            T->DW[i * PATCH_SZ + j] = G_desc[i] * G_desc[j];
}

/* surf.cpp SURFInvoker::operator() for one keypoint.  Sets kp->size = -1 on deletion. */
static void surf_describe_one(const uint8_t *img, int h, int w, int stride, const int32_t *sum,
                              const surf_tables *T, orc_keypoint *kp, float *vec, int extended, int upright)
{
    static const int dx_s[2][5] = { {0, 0, 2, 4, -1}, {2, 0, 4, 4, 1} };
    static const int dy_s[2][5] = { {0, 0, 4, 2, 1}, {0, 2, 4, 4, -1} };
    const int sw = w + 1, srows = h + 1, scols = w + 1;
    float X[169], Y[169], angle[169];
    uint8_t PATCH[PATCH_SZ + 1][PATCH_SZ + 1];
    float DX[PATCH_SZ][PATCH_SZ], DY[PATCH_SZ][PATCH_SZ];
    int dsize = extended ? 128 : 64;
    float size = kp->size;
    float cx = kp->x, cy = kp->y;
    float s = size * 1.2f / 9.0f;
    int grad_wav_size = 2 * cv_round_f(2 * s);
    if (srows < grad_wav_size || scols < grad_wav_size) { kp->size = -1; return; }
    float descriptor_dir = 360.f - 90.f;
    if (!upright) {
        surf_hf dx_t[2], dy_t[2];
        resize_haar(dx_s, dx_t, 2, 4, grad_wav_size, sw);
        resize_haar(dy_s, dy_t, 2, 4, grad_wav_size, sw);
        int nangle = 0;
        for (int kk = 0; kk < T->nOriSamples; kk++) {
            int x = cv_round_f(cx + T->aptx[kk] * s - (float)(grad_wav_size - 1) / 2);
            int y = cv_round_f(cy + T->apty[kk] * s - (float)(grad_wav_size - 1) / 2);
            if (y < 0 || y >= srows - grad_wav_size || x < 0 || x >= scols - grad_wav_size) continue;
            const int32_t *ptr = sum + (size_t)y * sw + x;
            float vx = calc_haar(ptr, dx_t, 2);
            float vy = calc_haar(ptr, dy_t, 2);
            X[nangle] = vx * T->aptw[kk];
            Y[nangle] = vy * T->aptw[kk];
            nangle++;
        }
        if (nangle == 0) { kp->size = -1; return; }
        for (int j = 0; j < nangle; j++) angle[j] = fast_atan2_deg(Y[j], X[j]); /* cv::phase(..., true) */
        float bestx = 0, besty = 0, descriptor_mod = 0;
        for (int i = 0; i < 360; i += SURF_ORI_SEARCH_INC) {
            float sumx = 0, sumy = 0, temp_mod;
            for (int j = 0; j < nangle; j++) {
                int d = abs(cv_round_f(angle[j]) - i);
                if (d < ORI_WIN / 2 || d > 360 - ORI_WIN / 2) { sumx += X[j]; sumy += Y[j]; }
            }
            temp_mod = sumx * sumx + sumy * sumy;
            if (temp_mod > descriptor_mod) { descriptor_mod = temp_mod; bestx = sumx; besty = sumy; }
        }
        descriptor_dir = fast_atan2_deg(-besty, bestx);
    }
    kp->angle = descriptor_dir;
    if (!vec) return;

    int win_size = (int)((PATCH_SZ + 1) * s);
    uint8_t *WIN = (uint8_t *)malloc((size_t)win_size * win_size);
    if (!upright) {
        descriptor_dir *= (float)(3.1415926535897932384626433832795 / 180);
        double sd, cd;
        det_sincos((double)descriptor_dir, &sd, &cd);   /* std::sin / std::cos on float: see det_sincos */
        float sin_dir = -(float)sd;
        float cos_dir = (float)cd;
        float win_offset = -(float)(win_size - 1) / 2;
        float start_x = cx + win_offset * cos_dir + win_offset * sin_dir;
        float start_y = cy - win_offset * sin_dir + win_offset * cos_dir;
        int ncols1 = w - 1, nrows1 = h - 1;
        for (int i = 0; i < win_size; i++, start_x += sin_dir, start_y += cos_dir) {
            double pixel_x = start_x;
            double pixel_y = start_y;
            for (int j = 0; j < win_size; j++, pixel_x += cos_dir, pixel_y -= sin_dir) {
                int ix = cv_floor_d(pixel_x), iy = cv_floor_d(pixel_y);
                if ((unsigned)ix < (unsigned)ncols1 && (unsigned)iy < (unsigned)nrows1) {
                    float a = (float)(pixel_x - ix), b = (float)(pixel_y - iy);
                    const uint8_t *p = img + (size_t)iy * stride + ix;
                    WIN[i * win_size + j] = (uint8_t)cv_round_f(p[0] * (1.f - a) * (1.f - b) + p[1] * a * (1.f - b) +
                                                                p[stride] * (1.f - a) * b + p[stride + 1] * a * b);
                } else {
                    int x = imin(imax(cv_round_d(pixel_x), 0), ncols1);
                    int y = imin(imax(cv_round_d(pixel_y), 0), nrows1);
                    WIN[i * win_size + j] = img[(size_t)y * stride + x];
                }
            }
        }
    } else {
        float win_offset = -(float)(win_size - 1) / 2;
        int start_x = cv_round_f(cx + win_offset);
        int start_y = cv_round_f(cy - win_offset);
        for (int i = 0; i < win_size; i++, start_x++) {
            int pixel_x = start_x, pixel_y = start_y;
            for (int j = 0; j < win_size; j++, pixel_y--) {
                int x = imax(pixel_x, 0), y = imax(pixel_y, 0);
                x = imin(x, w - 1); y = imin(y, h - 1);
                WIN[i * win_size + j] = img[(size_t)y * stride + x];
            }
        }
    }
    resize_area_u8(WIN, win_size, &PATCH[0][0]);
    free(WIN);

    for (int i = 0; i < PATCH_SZ; i++)
        for (int j = 0; j < PATCH_SZ; j++) {
            float dw = T->DW[i * PATCH_SZ + j];
            float vx = (PATCH[i][j + 1] - PATCH[i][j] + PATCH[i + 1][j + 1] - PATCH[i + 1][j]) * dw;
            float vy = (PATCH[i + 1][j] - PATCH[i][j] + PATCH[i + 1][j + 1] - PATCH[i][j + 1]) * dw;
            DX[i][j] = vx; DY[i][j] = vy;
        }
    for (int kk = 0; kk < dsize; kk++) vec[kk] = 0;
    double square_mag = 0;
    float *v = vec;
    if (extended) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                for (int y = i * 5; y < i * 5 + 5; y++)
                    for (int x = j * 5; x < j * 5 + 5; x++) {
                        float tx = DX[y][x], ty = DY[y][x];
                        if (ty >= 0) { v[0] += tx; v[1] += (float)fabs(tx); }
                        else { v[2] += tx; v[3] += (float)fabs(tx); }
                        if (tx >= 0) { v[4] += ty; v[5] += (float)fabs(ty); }
                        else { v[6] += ty; v[7] += (float)fabs(ty); }
                    }
                for (int kk = 0; kk < 8; kk++) square_mag += v[kk] * v[kk];
                v += 8;
            }
    } else {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                for (int y = i * 5; y < i * 5 + 5; y++)
                    for (int x = j * 5; x < j * 5 + 5; x++) {
                        float tx = DX[y][x], ty = DY[y][x];
                        v[0] += tx; v[1] += ty;
                        v[2] += (float)fabs(tx); v[3] += (float)fabs(ty);
                    }
                for (int kk = 0; kk < 4; kk++) square_mag += v[kk] * v[kk];
                v += 4;
            }
    }
    float scale = (float)(1. / (sqrt(square_mag) + DBL_EPSILON));
    for (int kk = 0; kk < dsize; kk++) vec[kk] *= scale;
}

int orc_surf_detect(const uint8_t *img, int h, int w, int stride,
                    double hessianThreshold, int nOctaves, int nOctaveLayers,
                    orc_keypoint *kps, int cap)
{
    int32_t *sum = (int32_t *)malloc(sizeof(int32_t) * (size_t)(h + 1) * (w + 1));
    orc_integral_u8_i32(img, h, w, stride, sum);
    kp_cand *c = NULL;
    int n = fast_hessian(sum, h, w, nOctaves, nOctaveLayers, (float)hessianThreshold, &c);
    free(sum);
    if (n > cap) { free(c); return -1; }
    for (int i = 0; i < n; i++) kps[i] = c[i].kp;
    free(c);
    return n;
}

int orc_surf_detect_describe(const uint8_t *img, int h, int w, int stride,
                             double hessianThreshold, int nOctaves, int nOctaveLayers,
                             int extended, int upright,
                             orc_keypoint *kps, float *desc, int cap, int nthreads)
{
    int dsize = extended ? 128 : 64;
    int32_t *sum = (int32_t *)malloc(sizeof(int32_t) * (size_t)(h + 1) * (w + 1));
    orc_integral_u8_i32(img, h, w, stride, sum);
    kp_cand *c = NULL;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    int n = fast_hessian(sum, h, w, nOctaves, nOctaveLayers, (float)hessianThreshold, &c);
    if (n > cap) { free(c); free(sum); return -1; }
    for (int i = 0; i < n; i++) kps[i] = c[i].kp;
    free(c);
    surf_tables T;
    surf_tables_init(&T);
#pragma omp parallel for schedule(dynamic, 16)
    for (int k = 0; k < n; k++)
        surf_describe_one(img, h, w, stride, sum, &T, &kps[k], desc ? desc + (size_t)k * dsize : NULL, extended, upright);
    free(sum);
    /* SURF_Impl::detectAndCompute: remove keypoints marked for deletion, preserving order */
    int j = 0;
    for (int i = 0; i < n; i++) {
        if (kps[i].size > 0) {
            if (i > j) {
                kps[j] = kps[i];
                if (desc) memcpy(desc + (size_t)j * dsize, desc + (size_t)i * dsize, sizeof(float) * dsize);
            }
            j++;
        }
    }
    return j;
}

/* ------------------------------------------------------------------------------------------
 * Brute-force matchers (features2d/src/matchers.cpp, core/src/batch_distance.cpp, stat.cpp)
 * ---------------------------------------------------------------------------------------- */
/* stat.cpp normL2Sqr_ scalar form: 4-wide groups accumulated into one float */
static inline float norm_l2_sqr(const float *a, const float *b, int n)
{
    int j = 0; float d = 0.f;
    for (; j <= n - 4; j += 4) {
        float t0 = a[j] - b[j], t1 = a[j + 1] - b[j + 1], t2 = a[j + 2] - b[j + 2], t3 = a[j + 3] - b[j + 3];
        d += t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3;
    }
    for (; j < n; j++) { float t = a[j] - b[j]; d += t * t; }
    return d;
}

void orc_bf_l2_knn2(const float *q, int nq, const float *t, int nt, int dim,
                    int32_t *idx1, float *d1, int32_t *idx2, float *d2, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nq; i++) {
        float b1 = INFINITY, b2 = INFINITY; int i1 = -1, i2 = -1;
        const float *a = q + (size_t)i * dim;
        for (int j = 0; j < nt; j++) {
            float d = sqrtf(norm_l2_sqr(a, t + (size_t)j * dim, dim)); /* batchDistL2_ */
            /* batch_distance.cpp K-insertion: if d < dist[K-1], shift larger entries */
            if (d < b1) { b2 = b1; i2 = i1; b1 = d; i1 = j; }
            else if (d < b2) { b2 = d; i2 = j; }
        }
        idx1[i] = i1; d1[i] = b1; idx2[i] = i2; d2[i] = b2;
    }
}

int orc_bf_l2_ratio_matches(const float *q, int nq, const float *t, int nt, int dim,
                            double ratio, int32_t *pairs, int nthreads)
{
    int32_t *i1 = (int32_t *)malloc(sizeof(int32_t) * (nq ? nq : 1)), *i2 = (int32_t *)malloc(sizeof(int32_t) * (nq ? nq : 1));
    float *d1 = (float *)malloc(sizeof(float) * (nq ? nq : 1)), *d2 = (float *)malloc(sizeof(float) * (nq ? nq : 1));
    orc_bf_l2_knn2(q, nq, t, nt, dim, i1, d1, i2, d2, nthreads);
    int m = 0;
    for (int i = 0; i < nq; i++) {
        /* ImageUtility.py:294: len(m)==2 and m[0].distance < m[1].distance * self.searchRatio  (Python floats) */
        if (i1[i] >= 0 && i2[i] >= 0 && (double)d1[i] < (double)d2[i] * ratio) {
            pairs[2 * m] = i1[i]; pairs[2 * m + 1] = i; m++;
        }
    }
    free(i1); free(i2); free(d1); free(d2);
    return m;
}

int orc_bf_hamming_matches(const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes,
                           int max_dist, int32_t *pairs, int32_t *dist_out)
{
    int m = 0;
    for (int i = 0; i < nq; i++) {
        int best = 0x7fffffff, bi = -1;
        for (int j = 0; j < nt; j++) {
            int d = 0;
            for (int k = 0; k < nbytes; k++) d += __builtin_popcount(q[(size_t)i * nbytes + k] ^ t[(size_t)j * nbytes + k]);
            if (d < best) { best = d; bi = j; }
        }
        if (bi < 0) continue;
        if (max_dist >= 0 && !(best < max_dist)) continue;
        pairs[2 * m] = bi; pairs[2 * m + 1] = i;
        if (dist_out) dist_out[m] = best;
        m++;
    }
    return m;
}

/* ------------------------------------------------------------------------------------------
 * Method.getOffsetByMode  (ImageUtility.py:139-178)
 * ---------------------------------------------------------------------------------------- */
void orc_mode_offset(const float *kpsA, const float *kpsB, const int32_t *pairs, int m,
                     int offsetEvaluate, int32_t *out4)
{
    out4[0] = 0; out4[1] = 0; out4[2] = 0; out4[3] = 0;
    if (m == 0) return;                                  /* :149-151 -> (False, [0,0]) */
    int32_t *dxs = (int32_t *)malloc(sizeof(int32_t) * m), *dys = (int32_t *)malloc(sizeof(int32_t) * m);
    int n = 0;
    for (int k = 0; k < m; k++) {
        int trainIdx = pairs[2 * k], queryIdx = pairs[2 * k + 1];
        /* ptA = (kpsA[q][1], kpsA[q][0]); float32 subtraction, int() truncates toward zero (:153-161) */
        float ax = kpsA[2 * queryIdx + 1], ay = kpsA[2 * queryIdx];
        float bx = kpsB[2 * trainIdx + 1], by = kpsB[2 * trainIdx];
        int dx = (int)(ax - bx), dy = (int)(ay - by);
        if (dx == 0 && dy == 0) continue;
        dxs[n] = dx; dys[n] = dy; n++;
    }
    if (n == 0) { dxs[0] = 0; dys[0] = 0; n = 1; }       /* :162-163 */
    /* mode of tuples, ties -> first inserted (dict insertion order + stable sort, :165-168) */
    int best = -1, bestcnt = 0;
    char *seen = (char *)calloc(n, 1);
    for (int i = 0; i < n; i++) {
        if (seen[i]) continue;
        int cnt = 0;
        for (int j = i; j < n; j++)
            if (dxs[j] == dxs[i] && dys[j] == dys[i]) { cnt++; seen[j] = 1; }
        if (cnt > bestcnt) { bestcnt = cnt; best = i; }
    }
    out4[0] = bestcnt >= offsetEvaluate;                 /* :175 */
    out4[1] = dxs[best]; out4[2] = dys[best]; out4[3] = bestcnt;
    free(dxs); free(dys); free(seen);
}

/* ------------------------------------------------------------------------------------------
 * cv2.phaseCorrelate  (imgproc/src/phasecorr.cpp, FP64, no window) -- Stitcher.py:230
 * Own mixed-radix complex FFT (radix 2/3/5 + generic) stands in for cv::dft; the rounding of the
 * transform itself is not part of the specification, the per-bin arithmetic around it is.
 * ---------------------------------------------------------------------------------------- */
typedef struct { double re, im; } cplx;

static void fft_rec(const cplx *in, cplx *out, int n, int stride, const cplx *tw, int twstride, cplx *scratch)
{
    if (n == 1) { out[0] = in[0]; return; }
    int p = (n % 2 == 0) ? 2 : (n % 3 == 0) ? 3 : (n % 5 == 0) ? 5 : n;
    int m = n / p;
    for (int r = 0; r < p; r++)
        fft_rec(in + (size_t)r * stride, out + (size_t)r * m, m, stride * p, tw, twstride * p, scratch);
    /* butterflies: X[k + q*m] = sum_r W_n^{r(k+q m)} Y_r[k] */
    for (int k = 0; k < m; k++) {
        cplx y[5]; cplx *yy = y;
        cplx *big = NULL;
        if (p > 5) { big = (cplx *)malloc(sizeof(cplx) * p); yy = big; }
        for (int r = 0; r < p; r++) {
            cplx v = out[(size_t)r * m + k];
            cplx t = tw[(size_t)((r * k) % n) * twstride];
            yy[r].re = v.re * t.re - v.im * t.im;
            yy[r].im = v.re * t.im + v.im * t.re;
        }
        for (int qd = 0; qd < p; qd++) {
            double sr = 0, si = 0;
            for (int r = 0; r < p; r++) {
                cplx t = tw[(size_t)((r * qd * m) % n) * twstride];
                sr += yy[r].re * t.re - yy[r].im * t.im;
                si += yy[r].re * t.im + yy[r].im * t.re;
            }
            scratch[qd].re = sr; scratch[qd].im = si;
        }
        for (int qd = 0; qd < p; qd++) out[(size_t)qd * m + k] = scratch[qd];
        if (big) free(big);
    }
}

/* in-place 1-D FFT of length n over data with element stride `stride`; sign -1 forward, +1 inverse */
static void fft_1d(cplx *data, int n, int stride, int sign, const cplx *tw_fwd, cplx *tmp_in, cplx *tmp_out)
{
    for (int i = 0; i < n; i++) {
        tmp_in[i] = data[(size_t)i * stride];
        if (sign > 0) tmp_in[i].im = -tmp_in[i].im;
    }
    cplx scratch[64];
    cplx *sc = scratch, *big = NULL;
    if (n > 64) { big = (cplx *)malloc(sizeof(cplx) * n); sc = big; }
    fft_rec(tmp_in, tmp_out, n, 1, tw_fwd, 1, sc);
    if (big) free(big);
    for (int i = 0; i < n; i++) {
        cplx v = tmp_out[i];
        if (sign > 0) v.im = -v.im;
        data[(size_t)i * stride] = v;
    }
}

static cplx *make_twiddles(int n)
{
    cplx *tw = (cplx *)malloc(sizeof(cplx) * n);
    for (int k = 0; k < n; k++) {
        double a = -2.0 * 3.14159265358979323846 * k / n;
        tw[k].re = cos(a); tw[k].im = sin(a);
    }
    return tw;
}

static void fft_2d(cplx *data, int M, int N, int sign)
{
    cplx *twN = make_twiddles(N), *twM = make_twiddles(M);
    int L = M > N ? M : N;
#pragma omp parallel
    {
        cplx *ti = (cplx *)malloc(sizeof(cplx) * L), *to = (cplx *)malloc(sizeof(cplx) * L);
#pragma omp for schedule(static)
        for (int y = 0; y < M; y++) fft_1d(data + (size_t)y * N, N, 1, sign, twN, ti, to);
#pragma omp for schedule(static)
        for (int x = 0; x < N; x++) fft_1d(data + x, M, N, sign, twM, ti, to);
        free(ti); free(to);
    }
    free(twN); free(twM);
}

void orc_phase_correlate_u8(const uint8_t *a, const uint8_t *b, int h, int w,
                            int strideA, int strideB, double *out3, int *MN)
{
    int M = orc_optimal_dft_size(h), N = orc_optimal_dft_size(w);
    if (MN) { MN[0] = M; MN[1] = N; }
    size_t sz = (size_t)M * N;
    cplx *F1 = (cplx *)calloc(sz, sizeof(cplx)), *F2 = (cplx *)calloc(sz, sizeof(cplx));
    /* copyMakeBorder(..., 0, M-rows, 0, N-cols, BORDER_CONSTANT, 0): zero pad bottom/right */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            F1[(size_t)y * N + x].re = (double)a[(size_t)y * strideA + x];
            F2[(size_t)y * N + x].re = (double)b[(size_t)y * strideB + x];
        }
    fft_2d(F1, M, N, -1);
    fft_2d(F2, M, N, -1);
    /* mulSpectrums(conjB) -> magSpectrums -> divSpectrums on the CCS-packed half spectrum.
     * Purely-real bins of the packed format ((u,v) with u in {0, M/2 if M even}, v in {0, N/2 if N even})
     * store x*x in magSpectrums, hence C = x/(x*x+eps) there; complex bins: C = P*m/(m*m+eps) with
     * m = sqrt(re^2+im^2) (imaginary slot of the magnitude array taken as 0). */
    const double eps = DBL_EPSILON;
    cplx *C = F1; /* reuse */
    for (int u = 0; u < M; u++)
        for (int v = 0; v < N; v++) {
            size_t k = (size_t)u * N + v;
            double pr = F1[k].re * F2[k].re + F1[k].im * F2[k].im;
            double pi = F1[k].im * F2[k].re - F1[k].re * F2[k].im;
            int real_u = (u == 0) || (M % 2 == 0 && u == M / 2);
            int real_v = (v == 0) || (N % 2 == 0 && v == N / 2);
            if (real_u && real_v) {
                double mg = pr * pr;
                C[k].re = pr / (mg + eps); C[k].im = 0;
            } else {
                double mg = sqrt(pr * pr + pi * pi);
                double denom = mg * mg + eps;
                C[k].re = (pr * mg) / denom;
                C[k].im = (pi * mg) / denom;
            }
        }
    fft_2d(C, M, N, +1); /* idft without DFT_SCALE */
    double *R = (double *)malloc(sizeof(double) * sz);
    for (size_t k = 0; k < sz; k++) R[k] = C[k].re;
    /* fftShift: swap quadrants of size (N>>1) x (M>>1); odd trailing row/col stays in place */
    int xMid = N >> 1, yMid = M >> 1;
    if (!(M == 1 && N == 1)) {
        if (xMid == 0 || yMid == 0) {
            /* 1-D case: not reachable from the stitcher (ROIs are >= 2x2) -- plain half swap */
            int n = M * N, mid = n >> 1;
            for (int i = 0; i < mid; i++) { double t = R[i]; R[i] = R[i + mid]; R[i + mid] = t; }
        } else {
            for (int y = 0; y < yMid; y++)
                for (int x = 0; x < xMid; x++) {
                    double t = R[(size_t)y * N + x]; R[(size_t)y * N + x] = R[(size_t)(y + yMid) * N + x + xMid];
                    R[(size_t)(y + yMid) * N + x + xMid] = t;
                    t = R[(size_t)y * N + x + xMid]; R[(size_t)y * N + x + xMid] = R[(size_t)(y + yMid) * N + x];
                    R[(size_t)(y + yMid) * N + x] = t;
                }
        }
    }
    /* minMaxLoc: first maximum in row-major order */
    size_t pk = 0; double best = R[0];
    for (size_t k = 1; k < sz; k++) if (R[k] > best) { best = R[k]; pk = k; }
    int py = (int)(pk / N), px = (int)(pk % N);
    /* weightedCentroid 5x5 clamped */
    int minr = py - 2, maxr = py + 2, minc = px - 2, maxc = px + 2;
    if (minr < 0) minr = 0;
    if (minc < 0) minc = 0;
    if (maxr > M - 1) maxr = M - 1;
    if (maxc > N - 1) maxc = N - 1;
    double cx = 0, cy = 0, s = 0;
    for (int y = minr; y <= maxr; y++)
        for (int x = minc; x <= maxc; x++) {
            double v = R[(size_t)y * N + x];
            cx += (double)x * v; cy += (double)y * v; s += v;
        }
    double response = s;
    s += DBL_EPSILON;
    cx /= s; cy /= s;
    response /= (double)M * N;   /* *response /= M*N  (int product promoted) */
    out3[0] = (double)N / 2.0 - cx;
    out3[1] = (double)M / 2.0 - cy;
    out3[2] = response;
    free(R); free(F1); free(F2);
}

/* ------------------------------------------------------------------------------------------
 * ImageFusion.fuseByFadeInAndFadeOut + getWeightsMatrix  (ImageFusion.py:192-244, 43-190)
 * ---------------------------------------------------------------------------------------- */
static inline int px_nonempty(const int64_t *A, int c, int ch, int i, int j)
{
    /* colour: imageA[i,j].sum() != -3 ; gray: imageA[i,j] != -1 */
    const int64_t *p = A + ((size_t)i * c + j) * ch;
    if (ch == 1) return p[0] != -1;
    int64_t s = 0;
    for (int k = 0; k < ch; k++) s += p[k];
    return s != -3;
}
static inline int pywrap(int i, int n) { return i < 0 ? i + n : i; }

/* Produces the separable float32 factors wB1[row], wB2[col] of weightMatB (= wB1*wB2), emulating the
 * Python loops including negative-index wrap.  Returns 0, or -1 where the reference would raise. */
static int corner_weights(const int64_t *A, int row, int col, int ch, float *wB1, float *wB2, int32_t *info)
{
    for (int i = 0; i < row; i++) wB1[i] = 1.f;
    for (int j = 0; j < col; j++) wB2[j] = 1.f;
    long cnt[4] = {0, 0, 0, 0};
    int r2 = row / 2, c2 = col / 2;
    for (int i = 0; i < row; i++)
        for (int j = 0; j < col; j++)
            for (int k = 0; k < ch; k++)
                if (A[((size_t)i * col + j) * ch + k] > 0) {
                    int q = (i < r2) ? (j < c2 ? 0 : 3) : (j < c2 ? 1 : 2);
                    cnt[q]++;
                }
    int index = 0;
    for (int q = 1; q < 4; q++) if (cnt[q] < cnt[index]) index = q;
    int rowIndex = 0, colIndex = 0;
    if (index == 2 || index == 3) {
        for (int j = 1; j < col; j++) {
            if (index == 2) { for (int i = row - 1; i >= 0; i--) if (px_nonempty(A, col, ch, i, col - j)) { rowIndex = i + 1; break; } }
            else            { for (int i = 0; i < row; i++)      if (px_nonempty(A, col, ch, i, col - j)) { rowIndex = i - 1; break; } }
            if (rowIndex != 0) break;
        }
        if (rowIndex >= row) return -1;                      /* IndexError in the reference */
        for (int i = col - 1; i >= 0; i--)
            if (px_nonempty(A, col, ch, pywrap(rowIndex, row), i)) { colIndex = i + 1; break; }
    } else {
        for (int j = 0; j < col; j++) {
            if (index == 0) { for (int i = 0; i < row; i++)      if (px_nonempty(A, col, ch, i, j)) { rowIndex = i - 1; break; } }
            else            { for (int i = row - 1; i >= 0; i--) if (px_nonempty(A, col, ch, i, j)) { rowIndex = i + 1; break; } }
            if (rowIndex != 0) break;
        }
        if (rowIndex >= row) return -1;
        for (int i = 0; i < col; i++)
            if (px_nonempty(A, col, ch, pywrap(rowIndex, row), i)) { colIndex = i - 1; break; }
    }
    if (info) { info[1] = index; info[2] = rowIndex; info[3] = colIndex; }
    /* row ramps */
    if (index == 2 || index == 1) {           /* for i in range(rowIndex+1): w[rowIndex-i] = (rowIndex-i)/rowIndex */
        int n = rowIndex + 1, ri = rowIndex;
        for (int i = 0; i < n; i++) {
            if (ri == 0) ri = 1;
            int idx = ri - i;
            if (idx >= row) return -1;
            wB1[pywrap(idx, row)] = (float)((double)(ri - i) * 1 / ri);
        }
    } else {                                  /* for i in range(rowIndex,row): w[i] = (row-i-1)/(row-rowIndex-1) */
        int ri = rowIndex;
        for (int i = rowIndex; i < row; i++) {
            if (ri == 0) ri = 1;
            if (row - ri - 1 == 0) return -1; /* ZeroDivisionError */
            wB1[pywrap(i, row)] = (float)((double)(row - i - 1) * 1 / (row - ri - 1));
        }
    }
    /* col ramps */
    if (index == 2 || index == 3) {           /* for i in range(colIndex+1): w[colIndex-i] = (colIndex-i)/colIndex */
        int n = colIndex + 1, ci = colIndex;
        for (int i = 0; i < n; i++) {
            if (ci == 0) ci = 1;
            int idx = ci - i;
            if (idx >= col) return -1;
            wB2[pywrap(idx, col)] = (float)((double)(ci - i) * 1 / ci);
        }
    } else {                                  /* for i in range(colIndex,col): w[i] = (col-i-1)/(col-colIndex-1) */
        int ci = colIndex;
        for (int i = colIndex; i < col; i++) {
            if (ci == 0) ci = 1;
            if (col - ci - 1 == 0) return -1;
            wB2[pywrap(i, col)] = (float)((double)(col - i - 1) * 1 / (col - ci - 1));
        }
    }
    return 0;
}

/* ImageFusion.getWeightsMatrix alone: separable factors of weightMatB.  Returns 0, -1 where the reference raises. */
int orc_corner_ramps(const int64_t *A, int r, int c, int ch, float *wB_r, float *wB_c, int32_t *info)
{
    return corner_weights(A, r, c, ch, wB_r, wB_c, info);
}

void orc_fuse_fade(int64_t *A, const int64_t *B, int r, int c, int ch, int dx, int dy,
                   uint8_t *out, int32_t *info)
{
    size_t npx = (size_t)r * c, nel = npx * ch;
    float *wA_r = (float *)malloc(sizeof(float) * r), *wB_r = (float *)malloc(sizeof(float) * r);
    float *wA_c = (float *)malloc(sizeof(float) * c), *wB_c = (float *)malloc(sizeof(float) * c);
    for (int i = 0; i < r; i++) wA_r[i] = wB_r[i] = 1.f;
    for (int j = 0; j < c; j++) wA_c[j] = wB_c[j] = 1.f;
    size_t valid = 0;
    for (size_t k = 0; k < nel; k++) valid += A[k] > -1;
    int corner = 0;
    if (info) { info[0] = 0; info[1] = -1; info[2] = 0; info[3] = 0; }
    if ((double)valid / (double)nel > 0.65) {
        if (c <= r) {
            for (int i = 0; i < c; i++) {
                if (dy >= 0) {   /* float32 array * int * 1.0 / int  -> float32 ops */
                    wA_c[c - i - 1] = ((wA_c[c - i - 1] * (float)i) * 1.0f) / (float)c;
                    wB_c[i] = ((wB_c[i] * (float)i) * 1.0f) / (float)c;
                } else {
                    wA_c[c - i - 1] = ((wA_c[c - i - 1] * (float)(c - i)) * 1.0f) / (float)c;
                    wB_c[i] = ((wB_c[i] * (float)(c - i)) * 1.0f) / (float)c;
                }
            }
        } else {
            for (int i = 0; i < r; i++) {
                if (dx <= 0) {
                    wA_r[i] = ((wA_r[i] * (float)i) * 1.0f) / (float)r;
                    wB_r[r - i - 1] = ((wB_r[r - i - 1] * (float)i) * 1.0f) / (float)r;
                } else {
                    wA_r[i] = ((wA_r[i] * (float)(r - i)) * 1.0f) / (float)r;
                    wB_r[r - i - 1] = ((wB_r[r - i - 1] * (float)(r - i)) * 1.0f) / (float)r;
                }
            }
        }
    } else {
        corner = 1;
        if (info) info[0] = 1;
        if (corner_weights(A, r, c, ch, wB_r, wB_c, info) != 0) {
            if (info) info[0] = -1;
            memset(out, 0, nel);
            free(wA_r); free(wB_r); free(wA_c); free(wB_c);
            return;
        }
    }
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) {
            float wA, wB;
            if (corner) { wB = wB_r[i] * wB_c[j]; wA = 1 - wB; }
            else { wA = wA_r[i] * wA_c[j]; wB = wB_r[i] * wB_c[j]; } /* one factor is exactly 1 */
            for (int k = 0; k < ch; k++) {
                size_t e = ((size_t)i * c + j) * ch + k;
                if (A[e] < 0) A[e] = B[e];                     /* imageA[imageA < 0] = imageB[imageA < 0] */
                double res = (double)wA * (double)A[e] + (double)wB * (double)B[e];
                if (res < 0) res = 0;
                if (res > 255) res = 255;
                out[e] = (uint8_t)res;                         /* np.uint8(): truncation */
            }
        }
    free(wA_r); free(wB_r); free(wA_c); free(wB_c);
}

void orc_det_sincos(const double *x, int n, double *out)
{
    for (int i = 0; i < n; i++) det_sincos(x[i], &out[2 * i], &out[2 * i + 1]);
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * Pre-enhancement of Stitcher.py:269-276 / 327-334 (Method.isEnhance): cv2.equalizeHist and cv2.createCLAHE(clipLimit,
 * (tileSize, tileSize)).apply on 8-bit images -- OpenCV 3.3.1 imgproc/src/histogram.cpp (equalizeHist) and clahe.cpp, restated
 * from the published algorithm (sources not under /root/reference: parity unpinned against cv2).
 * ------------------------------------------------------------------------------------------------------------------------- */
static inline uint8_t sat_u8_f(float v) { int iv = (int)lrintf(v); return (uint8_t)(iv < 0 ? 0 : iv > 255 ? 255 : iv); }

/* equalizeHist: hist -> first non-empty bin i0 -> lut[i] = saturate(round((sum_{i0 < j <= i} hist[j]) * 255.f / (total - hist[i0]))) */
void orc_equalize_hist(const uint8_t *src, int h, int w, int stride, uint8_t *dst)
{
    int hist[256] = {0};
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) hist[src[(size_t)y * stride + x]]++;
    int i = 0;
    while (!hist[i]) ++i;
    const int total = h * w;
    if (hist[i] == total) { for (int y = 0; y < h; y++) memset(dst + (size_t)y * w, i, (size_t)w); return; }
    const float scale = (256 - 1.f) / (total - hist[i]);
    uint8_t lut[256];
    memset(lut, 0, sizeof(lut));
    int sum = 0;
    for (lut[i++] = 0; i < 256; ++i) { sum += hist[i]; lut[i] = sat_u8_f(sum * scale); }
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) dst[(size_t)y * w + x] = lut[src[(size_t)y * stride + x]];
}

static inline int reflect101_i(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

/* CLAHE_Impl::apply for CV_8UC1: tiles x tiles grid (image extended bottom / right with BORDER_REFLECT_101 to a multiple of the
 * grid when it is not one), per-tile clipped histogram -> LUT, bilinear interpolation of the four neighbouring tile LUTs. */
void orc_clahe(const uint8_t *src, int h, int w, int stride, double clipLimitD, int tilesX, int tilesY, uint8_t *dst)
{
    const int histSize = 256;
    int eh = h, ew = w;
    if (!(w % tilesX == 0 && h % tilesY == 0)) { eh = h + (tilesY - (h % tilesY)); ew = w + (tilesX - (w % tilesX)); }
    const int tw = ew / tilesX, th = eh / tilesY;
    const int tileSizeTotal = tw * th;
    const float lutScale = (float)(histSize - 1) / tileSizeTotal;
    int clipLimit = 0;
    if (clipLimitD > 0.0) {
        clipLimit = (int)(clipLimitD * tileSizeTotal / histSize);
        if (clipLimit < 1) clipLimit = 1;
    }
    uint8_t *lut = (uint8_t *)malloc((size_t)tilesX * tilesY * histSize);
    for (int k = 0; k < tilesX * tilesY; k++) {
        const int ty = k / tilesX, tx = k % tilesX;
        int tileHist[256] = {0};
        for (int y = ty * th; y < (ty + 1) * th; y++)
            for (int x = tx * tw; x < (tx + 1) * tw; x++)
                tileHist[src[(size_t)reflect101_i(y, h) * stride + reflect101_i(x, w)]]++;
        if (clipLimit > 0) {
            int clipped = 0;
            for (int i = 0; i < histSize; ++i)
                if (tileHist[i] > clipLimit) { clipped += tileHist[i] - clipLimit; tileHist[i] = clipLimit; }
            const int redistBatch = clipped / histSize;
            const int residual = clipped - redistBatch * histSize;
            for (int i = 0; i < histSize; ++i) tileHist[i] += redistBatch;
            for (int i = 0; i < residual; ++i) tileHist[i]++;          /* 3.3.1: the first `residual` bins (3.4.2+ strides them) */
        }
        int sum = 0;
        uint8_t *tileLut = lut + (size_t)k * histSize;
        for (int i = 0; i < histSize; ++i) { sum += tileHist[i]; tileLut[i] = sat_u8_f(sum * lutScale); }
    }
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    for (int y = 0; y < h; y++) {
        const float tyf = y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tilesY - 1) ty2 = tilesY - 1;
        const uint8_t *lutPlane1 = lut + (size_t)ty1 * tilesX * histSize, *lutPlane2 = lut + (size_t)ty2 * tilesX * histSize;
        for (int x = 0; x < w; x++) {
            const float txf = x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            const float xa = txf - tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tilesX - 1) tx2 = tilesX - 1;
            const int srcVal = src[(size_t)y * stride + x];
            const int ind1 = tx1 * histSize + srcVal, ind2 = tx2 * histSize + srcVal;
            const float res = (lutPlane1[ind1] * xa1 + lutPlane1[ind2] * xa) * ya1 + (lutPlane2[ind1] * xa1 + lutPlane2[ind2] * xa) * ya;
            dst[(size_t)y * w + x] = sat_u8_f(res);
        }
    }
    free(lut);
}

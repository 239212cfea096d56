/*
 * vfsms_oracle_orb.c -- CPU ORACLE for ORB detect+describe (test infrastructure only; see vfsms_oracle.h).
 *
 * Restates cv2.ORB_create(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, WTA_K=2, HARRIS_SCORE, patchSize,
 * fastThreshold).detectAndCompute(image, None) as called at ImageUtility.py:260,262 -- OpenCV 3.3.1
 * modules/features2d/src/{orb,fast,fast_score,keypoint}.cpp and the imgproc pieces it drives
 * (resize INTER_LINEAR 8U fixed point, copyMakeBorder REFLECT_101, GaussianBlur 7x7 sigma 2 as the 8-bit fixed-point
 * separable filter).  Sources are not under /root/reference (un-vendored opencv-python 3.3.1.11): PARITY UNPINNED
 * against OpenCV itself; pinned on synthetic grids with exact integer ground truth.
 *
 * Sampling pattern: patchSize 31 uses upstream's learned 256x4 table bit_pattern_31_ (restated in
 * imagestitch_amd/csrc/orb_pattern31.h, structurally checked, unverifiable against cv2 offline); other patch sizes use
 * upstream's makeRandomPattern (RNG(0x34985739), MWC).
 *
 * One stated deviation from upstream, because the upstream artefact cannot be reproduced offline:
 *   KeyPointsFilter::retainBest keeps every keypoint whose response >= the n-th best, in detection (row-major scan)
 *   order.  Upstream calls std::nth_element and then reads keypoints[n-1].response as the tie threshold: both the order
 *   it leaves behind and, when integer FAST scores tie at the cut, which of the tied keypoints survive depend on the C++
 *   standard library the wheel was built with (MSVC's for opencv-python 3.3.1.11 on the reference's Windows host).
 */
#include "vfsms_oracle.h"
#include "../imagestitch_amd/csrc/orb_pattern31.h"   /* the table is DATA shared with the engine */
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil_d(double v) { int i = (int)v; return i + (i < v); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

typedef struct { uint8_t *data; int h, w; } level_img;   /* contiguous, stride == w */

/* imgproc/src/resize.cpp: resize(8U, INTER_LINEAR): 11-bit fixed-point coefficients, horizontal pass in int,
 * vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2 */
static void resize_linear_u8(const level_img *src, level_img *dst)
{
    const int sw = src->w, sh = src->h, dw = dst->w, dh = dst->h;
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_d(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) { xmax = imin(xmax, dx); if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
        xofs[dx] = sx;
        ialpha[2 * dx] = (short)cv_round_f((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = (short)cv_round_f(fx * 2048);
    }
    int *row0 = (int *)malloc(sizeof(int) * dw), *row1 = (int *)malloc(sizeof(int) * dw);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_d(fy);
        fy -= sy;
        short b0 = (short)cv_round_f((1.f - fy) * 2048), b1 = (short)cv_round_f(fy * 2048);
        int sy0 = imin(imax(sy, 0), sh - 1), sy1 = imin(imax(sy + 1, 0), sh - 1);
        const uint8_t *S0 = src->data + (size_t)sy0 * sw, *S1 = src->data + (size_t)sy1 * sw;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            if (dx < xmax) {
                row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx + 1] * ialpha[2 * dx + 1];
                row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx + 1] * ialpha[2 * dx + 1];
            } else {
                row0[dx] = S0[sx] * 2048;
                row1[dx] = S1[sx] * 2048;
            }
        }
        uint8_t *D = dst->data + (size_t)dy * dw;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(row0); free(row1);
}

/* features2d/src/fast_score.cpp cornerScore<16> */
static int corner_score16(const uint8_t *ptr, const int *pixel, int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = imin((int)d[k + 1], (int)d[k + 2]);
        a = imin(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = imin(a, (int)d[k + 4]); a = imin(a, (int)d[k + 5]); a = imin(a, (int)d[k + 6]);
        a = imin(a, (int)d[k + 7]); a = imin(a, (int)d[k + 8]);
        a0 = imax(a0, imin(a, (int)d[k]));
        a0 = imax(a0, imin(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = imax((int)d[k + 1], (int)d[k + 2]);
        b = imax(b, (int)d[k + 3]); b = imax(b, (int)d[k + 4]); b = imax(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = imax(b, (int)d[k + 6]); b = imax(b, (int)d[k + 7]); b = imax(b, (int)d[k + 8]);
        b0 = imin(b0, imax(b, (int)d[k]));
        b0 = imin(b0, imax(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

typedef struct { float x, y, response, angle; int level; } orb_kp;

/* features2d/src/fast.cpp FAST_t<16>(img, keypoints, threshold, nonmax_suppression = true): score map + 3x3 NMS,
 * keypoints in row-major order */
static int fast16(const level_img *img, int threshold, orb_kp **out)
{
    static const int off[16][2] = { {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3} };
    const int h = img->h, w = img->w;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = off[k][0] + off[k][1] * w;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = imin(imax(threshold, 0), 255);
    uint8_t *score = (uint8_t *)calloc((size_t)h * w, 1);
    for (int i = 3; i < h - 3; i++)
        for (int j = 3; j < w - 3; j++) {
            const uint8_t *ptr = img->data + (size_t)i * w + j;
            int v = ptr[0];
            int is_corner = 0;
            for (int pass = 0; pass < 2 && !is_corner; pass++) {
                int count = 0;
                for (int k = 0; k < 25; k++) {
                    int x = ptr[pixel[k]];
                    int hit = pass == 0 ? (x < v - threshold) : (x > v + threshold);
                    if (hit) { if (++count > 8) { is_corner = 1; break; } }
                    else count = 0;
                }
            }
            if (is_corner) score[(size_t)i * w + j] = (uint8_t)corner_score16(ptr, pixel, threshold);
        }
    int cap = 1024, n = 0;
    orb_kp *kp = (orb_kp *)malloc(sizeof(orb_kp) * cap);
    for (int i = 3; i < h - 3; i++)
        for (int j = 3; j < w - 3; j++) {
            int s = score[(size_t)i * w + j];
            if (!s) continue;
            const uint8_t *p = score + (size_t)i * w + j;
            if (s > p[1] && s > p[-1] && s > p[-w - 1] && s > p[-w] && s > p[-w + 1] && s > p[w - 1] && s > p[w] && s > p[w + 1]) {
                if (n == cap) { cap *= 2; kp = (orb_kp *)realloc(kp, sizeof(orb_kp) * cap); }
                kp[n].x = (float)j; kp[n].y = (float)i; kp[n].response = (float)s; kp[n].angle = -1; kp[n].level = 0; n++;
            }
        }
    free(score);
    *out = kp;
    return n;
}

static int cmp_float_desc(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return x > y ? -1 : x < y ? 1 : 0;
}
/* KeyPointsFilter::retainBest: keep every keypoint with response >= the n-th best (detection order preserved) */
static int retain_best(orb_kp *kp, int n, int n_points)
{
    if (n_points < 0 || n <= n_points) return n;
    if (n_points == 0) return 0;
    float *r = (float *)malloc(sizeof(float) * n);
    for (int i = 0; i < n; i++) r[i] = kp[i].response;
    qsort(r, n, sizeof(float), cmp_float_desc);
    float T = r[n_points - 1];
    free(r);
    int m = 0;
    for (int i = 0; i < n; i++) if (kp[i].response >= T) kp[m++] = kp[i];
    return m;
}

/* orb.cpp HarrisResponses (blockSize 7, k = 0.04) on the un-blurred level image */
static float harris_response(const level_img *img, int x0, int y0)
{
    const int step = img->w, r = 3, bs = 7;
    float scale = 1.f / ((1 << 2) * bs * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    const uint8_t *ptr0 = img->data + (size_t)(y0 - r) * step + x0 - r;
    int a = 0, b = 0, c = 0;
    for (int k = 0; k < bs * bs; k++) {
        const uint8_t *ptr = ptr0 + (k / bs) * step + (k % bs);
        int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
        int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
        a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
    }
    return ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

static inline float fast_atan2_deg(float y, float x)
{
    const float s = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s, p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* imgproc GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on 8U: separable filter with 8-bit fixed-point coefficients
 * (cvRound(k*256)), row pass in int, column pass (sum + 2^15) >> 16 */
static void gaussian_blur7_u8(const level_img *src, level_img *dst)
{
    int kf[7];
    {
        float cf[7]; double sum = 0; const double sigma = 2.0, scale2X = -0.5 / (sigma * sigma);
        for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)exp(scale2X * x * x); sum += cf[i]; }
        sum = 1. / sum;
        for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); kf[i] = cv_round_f(cf[i] * 256.f); }
    }
    const int h = src->h, w = src->w;
    int *tmp = (int *)malloc(sizeof(int) * (size_t)h * w);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -3; k <= 3; k++) s += kf[k + 3] * src->data[(size_t)y * w + reflect101(x + k, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -3; k <= 3; k++) s += kf[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
            s = (s + (1 << 15)) >> 16;
            dst->data[(size_t)y * w + x] = (uint8_t)(s < 0 ? 0 : s > 255 ? 255 : s);
        }
    free(tmp);
}

/* orb.cpp makeRandomPattern: RNG rng(0x34985739); x, y = rng.uniform(-patchSize/2, patchSize/2+1) */
void orc_orb_pattern(int patchSize, int npoints, int32_t *xy)
{
    if (patchSize == 31) {          /* ORB_Impl::detectAndCompute: pattern0 = bit_pattern_31_ unless patchSize != 31 */
        for (int i = 0; i < 2 * npoints && i < 1024; i++) xy[i] = VFSMS_ORB_BIT_PATTERN_31[i];
        return;
    }
    uint64_t state = 0x34985739ULL;
    for (int i = 0; i < 2 * npoints; i++) {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        unsigned r = (unsigned)state;
        int a = -patchSize / 2, b = patchSize / 2 + 1;
        xy[i] = (int)(r % (unsigned)(b - a) + a);
    }
}

/* ORB_Impl::detectAndCompute.  kps: orc_keypoint (x, y in level-0 coordinates, size, angle, response, octave);
 * desc: uint8 [cap][32].  Returns n or -1 on capacity overflow. */
int orc_orb_detect_describe(const uint8_t *image, int h, int w, int stride,
                            int nfeatures, float scaleFactorF, int nlevels, int edgeThreshold, int firstLevel,
                            int patchSize, int fastThreshold,
                            orc_keypoint *kps, uint8_t *desc, int cap)
{
    const double scaleFactor = (double)scaleFactorF;
    const int halfPatchSize = patchSize / 2;
    level_img *lv = (level_img *)calloc(nlevels, sizeof(level_img));
    float *layerScale = (float *)malloc(sizeof(float) * nlevels);
    for (int l = 0; l < nlevels; l++) {
        float scale = (float)pow(scaleFactor, (double)(l - firstLevel));
        layerScale[l] = scale;
        lv[l].w = cv_round_f(w / scale); lv[l].h = cv_round_f(h / scale);
        lv[l].data = (uint8_t *)malloc((size_t)imax(lv[l].w, 1) * imax(lv[l].h, 1));
    }
    for (int l = 0; l < nlevels; l++) {
        if (l == firstLevel) { for (int y = 0; y < h; y++) memcpy(lv[l].data + (size_t)y * w, image + (size_t)y * stride, w); }
        else {
            level_img prev;
            if (l > firstLevel && l > 0) prev = lv[l - 1];
            else { prev.data = (uint8_t *)malloc((size_t)h * w); prev.h = h; prev.w = w; for (int y = 0; y < h; y++) memcpy(prev.data + (size_t)y * w, image + (size_t)y * stride, w); }
            if (lv[l].w > 0 && lv[l].h > 0) resize_linear_u8(&prev, &lv[l]);
            if (!(l > firstLevel && l > 0)) free(prev.data);
        }
    }
    /* computeKeyPoints */
    int *nfeat = (int *)malloc(sizeof(int) * nlevels);
    {
        float factor = (float)(1.0 / scaleFactor);
        float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int l = 0; l < nlevels - 1; l++) { nfeat[l] = cv_round_f(nd); sum += nfeat[l]; nd *= factor; }
        nfeat[nlevels - 1] = imax(nfeatures - sum, 0);
    }
    int *umax = (int *)calloc(halfPatchSize + 2, sizeof(int));
    {
        int v, v0, vmax = cv_floor_d(halfPatchSize * sqrtf(2.f) / 2 + 1);
        int vmin = cv_ceil_d(halfPatchSize * sqrtf(2.f) / 2);
        for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt((double)halfPatchSize * halfPatchSize - v * v));
        for (v = halfPatchSize, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }
    int total = 0, allcap = 4096;
    orb_kp *all = (orb_kp *)malloc(sizeof(orb_kp) * allcap);
    for (int l = 0; l < nlevels; l++) {
        orb_kp *kp = NULL;
        int n = 0;
        if (lv[l].w > 6 && lv[l].h > 6) n = fast16(&lv[l], fastThreshold, &kp);
        /* runByImageBorder(edgeThreshold): Rect(border, border, w-2*border, h-2*border).contains(pt) */
        int m = 0;
        for (int i = 0; i < n; i++)
            if (kp[i].x >= edgeThreshold && kp[i].x < lv[l].w - edgeThreshold && kp[i].y >= edgeThreshold && kp[i].y < lv[l].h - edgeThreshold)
                kp[m++] = kp[i];
        n = retain_best(kp, m, 2 * nfeat[l]);                 /* HARRIS_SCORE: keep 2x, re-rank below */
        for (int i = 0; i < n; i++) {
            kp[i].level = l;
            kp[i].response = harris_response(&lv[l], cv_round_f(kp[i].x), cv_round_f(kp[i].y));
        }
        n = retain_best(kp, n, nfeat[l]);
        for (int i = 0; i < n; i++) {
            /* ICAngles */
            const uint8_t *center = lv[l].data + (size_t)cv_round_f(kp[i].y) * lv[l].w + cv_round_f(kp[i].x);
            const int step = lv[l].w;
            int m_01 = 0, m_10 = 0;
            for (int u = -halfPatchSize; u <= halfPatchSize; ++u) m_10 += u * center[u];
            for (int v = 1; v <= halfPatchSize; ++v) {
                int v_sum = 0, d = umax[v];
                for (int u = -d; u <= d; ++u) {
                    int val_plus = center[u + v * step], val_minus = center[u - v * step];
                    v_sum += (val_plus - val_minus);
                    m_10 += u * (val_plus + val_minus);
                }
                m_01 += v * v_sum;
            }
            kp[i].angle = fast_atan2_deg((float)m_01, (float)m_10);
            if (total == allcap) { allcap *= 2; all = (orb_kp *)realloc(all, sizeof(orb_kp) * allcap); }
            all[total++] = kp[i];
        }
        free(kp);
    }
    int ret = total;
    if (total > cap) ret = -1;
    else if (total > 0) {
        int32_t *pattern = (int32_t *)malloc(sizeof(int32_t) * 2 * 512);
        orc_orb_pattern(patchSize, 512, pattern);
        level_img *bl = (level_img *)calloc(nlevels, sizeof(level_img));
        for (int l = 0; l < nlevels; l++) {
            bl[l] = lv[l];
            bl[l].data = (uint8_t *)malloc((size_t)imax(lv[l].w, 1) * imax(lv[l].h, 1));
            if (lv[l].w > 0 && lv[l].h > 0) gaussian_blur7_u8(&lv[l], &bl[l]);
        }
        for (int j = 0; j < total; j++) {
            const orb_kp *k = &all[j];
            const float sf = layerScale[k->level];
            /* computeKeyPoints tail: pt *= scale;  computeOrbDescriptors: cvRound(pt * (1.f/scale)) */
            float px = k->x * sf, py = k->y * sf;
            kps[j].x = px; kps[j].y = py; kps[j].size = patchSize * sf; kps[j].angle = k->angle;
            kps[j].response = k->response; kps[j].octave = k->level; kps[j].class_id = -1;
            float inv = 1.f / sf;
            float angle = k->angle * (float)(3.1415926535897932384626433832795 / 180.f);
            double sd, cd;
            det_sincos((double)angle, &sd, &cd);
            float a = (float)cd, b = (float)sd;
            const level_img *L = &bl[k->level];
            const uint8_t *center = L->data + (size_t)cv_round_f(py * inv) * L->w + cv_round_f(px * inv);
            const int step = L->w;
            uint8_t *d = desc + (size_t)j * 32;
            const int32_t *pat = pattern;
            for (int i = 0; i < 32; ++i, pat += 32) {
                int val = 0;
                for (int bit = 0; bit < 8; bit++) {
                    int p0x = pat[4 * bit], p0y = pat[4 * bit + 1], p1x = pat[4 * bit + 2], p1y = pat[4 * bit + 3];
                    float x0 = p0x * a - p0y * b, y0 = p0x * b + p0y * a;
                    float x1 = p1x * a - p1y * b, y1 = p1x * b + p1y * a;
                    int t0 = center[cv_round_f(y0) * step + cv_round_f(x0)];
                    int t1 = center[cv_round_f(y1) * step + cv_round_f(x1)];
                    val |= (t0 < t1) << bit;
                }
                d[i] = (uint8_t)val;
            }
        }
        for (int l = 0; l < nlevels; l++) free(bl[l].data);
        free(bl); free(pattern);
    }
    for (int l = 0; l < nlevels; l++) free(lv[l].data);
    free(lv); free(layerScale); free(nfeat); free(umax); free(all);
    return ret;
}

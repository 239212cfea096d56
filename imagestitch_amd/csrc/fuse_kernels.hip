// fuse_kernels.hip -- device-resident mosaic canvas + fadeInAndFadeOut blend for gfx950.
//
// Replaces the paste/fuse loop of Stitcher.getStitchByOffset (Stitcher.py:440-483) and
// ImageFusion.fuseByFadeInAndFadeOut / getWeightsMatrix (ImageFusion.py:192-244, 43-190).
// The canvas is u8 + a validity plane (the reference keeps int64 with -1 = empty, Stitcher.py:434-436;
// 8x-24x the bytes).  Weight matrices are never materialised: the blend kernel takes the separable
// float32 ramps (wA_r, wA_c, wB_r, wB_c: a few KB) and forms wA, wB per pixel exactly as numpy does
// (float32 weights, float64 multiply-add, clamp, truncate).  The statistics the reference gathers with
// Python pixel loops (occupancy ratio, quadrant counts, first non-empty pixel scans) come from one
// reduction kernel; the ramp construction itself (a few hundred scalar ops with the reference's index
// quirks) runs on the host.
#include "common.h"
#include "detmath.h"
#include <vector>
#include <algorithm>
#include <string.h>
#include <stdlib.h>

struct FuseStats {                  // device-side result of k_fuse_stats
    unsigned long long valid;       // count_nonzero(A > -1)  (elements)
    unsigned long long quad[4];     // count_nonzero(A[quadrant] > 0): TL, BL, BR, TR (elements)
};

// accessors ------------------------------------------------------------------------------------------
struct CanvasView {                 // A = canvas region, B = tile
    const uint8_t *pix; const uint8_t *mask; int ccols, ch; int ry0, rx0;
    const uint8_t *tile; int tw; int ty0, tx0;     // ROI origin inside the tile
    __device__ bool a_valid(int i, int j) const { return mask[(size_t)(ry0 + i) * ccols + rx0 + j] != 0; }
    __device__ int a_val(int i, int j, int k) const { return pix[((size_t)(ry0 + i) * ccols + rx0 + j) * ch + k]; }
    __device__ int b_val(int i, int j, int k) const { return tile[((size_t)(ty0 + i) * tw + tx0 + j) * ch + k]; }
};
struct I64View {
    const long long *A; const long long *B; int c, ch;
    __device__ bool a_valid(int i, int j) const {
        const long long *p = A + ((size_t)i * c + j) * ch;
        if (ch == 1) return p[0] != -1;
        long long s = 0; for (int k = 0; k < ch; k++) s += p[k];
        return s != -3;
    }
    __device__ long long a_raw(int i, int j, int k) const { return A[((size_t)i * c + j) * ch + k]; }
    __device__ long long b_val(int i, int j, int k) const { return B[((size_t)i * c + j) * ch + k]; }
};

typedef uint32_t u32u1 __attribute__((aligned(1)));

// the blend: writes the whole tile rectangle (outside the ROI: plain paste) and marks it valid
__device__ __forceinline__ uint8_t fade_px(float wA, float wB, bool av, int a0, int b)
{
    const int a = av ? a0 : b;                                         // imageA[imageA < 0] = imageB[imageA < 0]
    double res = (double)wA * (double)a + (double)wB * (double)b;
    res = res < 0 ? 0 : res;
    res = res > 255 ? 255 : res;
    return (uint8_t)res;                                               // np.uint8(): truncation
}
// ImageFusion.fuseByTrigonometric (ImageFusion.py:246-293): the same regions and geometry decisions as the fade, other weights:
// wA = sin(w pi / 2)^2, wB = 1 - wA with w the float64 ramp i / n or (n - i) / n of the strip modes, or getWeightsMatrix's float32
// 1 - wB1 wB2 in corner mode (numpy keeps float32 through `* math.pi / 2`, np.sin, np.power(., 2) and `1 -`).  sin is the explicit
// double-precision algorithm of detmath.h (numpy's own SIMD sin may differ from it in the last ulp: tests allow one grey level on
// < 0.1 % of the bytes).
struct TrigGeom { int on, r, c, dx, dy; };
__device__ __forceinline__ uint8_t trig_px(const TrigGeom &G, bool corner, int i, int j, float wbr, float wbc, bool av, int a0, int b)
{
    double wA, wB;
    if (corner) {
        const float wBf = wbr * wbc, wAf = 1.f - wBf;
        const float x = (wAf * 3.14159274101257324f) / 2;
        double sd, cd;
        det_sincos((double)x, &sd, &cd);
        const float s = (float)sd, wa = s * s, wb = 1.f - wa;
        wA = (double)wa; wB = (double)wb;
    } else {
        const double t = G.c <= G.r ? (double)(G.dy >= 0 ? j : G.c - j) / (double)G.c : (double)(G.dx <= 0 ? i : G.r - i) / (double)G.r;
        double sd, cd;
        det_sincos((t * 3.141592653589793) / 2, &sd, &cd);
        wA = sd * sd; wB = 1 - wA;
    }
    const int a = av ? a0 : b;                                         // imageA[imageA < 0] = imageB[imageA < 0]
    double res = wA * (double)a + wB * (double)b;
    res = res < 0 ? 0 : res;
    res = res > 255 ? 255 : res;
    return (uint8_t)res;
}
// the strip ramps of fuseByFadeInAndFadeOut in closed form, with the reference's float32 expression ((1. * f) * 1.0) / n:
//   col <= row: weightMatA_2[col - i - 1] = weightMatB_2[i] = f(i) / col, f(i) = i (dy >= 0) or col - i;  else  weightMatA_1[i] =
//   weightMatB_1[row - i - 1] = g(i) / row, g(i) = i (dx <= 0) or row - i  (what fuse_weights_body's strip branch stores into the arrays)
// kind 3: getWeightsMatrix's corner ramps (ImageFusion.py:43-190 as fuse_weights_body stores them) from (index, rowIndex, colIndex):
//   rows, index 2 / 1: weightMatB_1[i] = i / ri for 0 <= i <= rowIndex (ri = rowIndex, 0 patched to 1: then only [1] = 1 is written);
//         index 3 / 0: weightMatB_1[i] = (row - i - 1) / (row - ri - 1) for i >= max(rowIndex, 0);   columns alike with colIndex, index 2 / 3 | 0 / 1
//   quotients in float64, stored as float32; everything else stays 1
struct AnalyticRamps {
    int kind, r, c, dx, dy;
    int index, rowIndex, colIndex;
    __device__ __forceinline__ float corner_b(int i, int n, int at, bool counting_up) const
    {
        const int ai = at == 0 ? 1 : at;
        if (counting_up) return (at >= 1 && i <= at) ? (float)((double)i * 1 / ai) : 1.f;
        return i >= max(at, 0) ? (float)((double)(n - i - 1) * 1 / (n - ai - 1)) : 1.f;
    }
    __device__ __forceinline__ float cb_row(int i) const { return corner_b(i, r, rowIndex, index == 2 || index == 1); }
    __device__ __forceinline__ float cb_col(int j) const { return corner_b(j, c, colIndex, index == 2 || index == 3); }
    __device__ __forceinline__ float ratio(int n, int d) const { return ((1.f * (float)n) * 1.0f) / (float)d; }
    __device__ __forceinline__ float a_col(int j) const { return kind == 1 ? ratio(dy >= 0 ? c - 1 - j : j + 1, c) : 1.f; }
    __device__ __forceinline__ float b_col(int j) const { return kind == 1 ? ratio(dy >= 0 ? j : c - j, c) : 1.f; }
    __device__ __forceinline__ float a_row(int i) const { return kind == 2 ? ratio(dx <= 0 ? i : r - i, r) : 1.f; }
    __device__ __forceinline__ float b_row(int i) const { return kind == 2 ? ratio(dx <= 0 ? r - 1 - i : i + 1, r) : 1.f; }
};
// (bx, by): the block of the tile this workgroup blends -- blockIdx for the per-tile launch, a drawn index inside k_mosaic_walk
__device__ __forceinline__ void fuse_apply_block(uint8_t *pix, uint8_t *mask, int ccols, int ch,
                                                 const uint8_t *tile, int th, int tw, int y0, int x0,
                                                 int ry0, int rx0, int r, int c, const int *mode,
                                                 const float *wAr, const float *wAc, const float *wBr, const float *wBc, const TrigGeom &TG,
                                                 unsigned bx, unsigned by, int analytic = 0)
{
    const int y = (int)by;
    const int cy = y0 + y;
    const int i = cy - ry0;
    const bool row_in = i >= 0 && i < r;
    const int corner = analytic == 3 ? 1 : analytic ? 0 : mode[0];
    // analytic: the ROI is a strip (more than 65 % of it valid -- counted by the host from the rectangles placed so far) and the ramps are the
    // closed forms of fuseByFadeInAndFadeOut's strip branch (ImageFusion.py:206-221), formed here instead of read from the statistics kernel's
    // arrays: 1 = ramps along the columns (col <= row), 2 = along the rows; every other weight is 1.f as the reference initialises them
    AnalyticRamps AR = {analytic, r, c, TG.dx, TG.dy, 0, 0, 0};
    if (analytic == 3) { AR.index = mode[2]; AR.rowIndex = mode[3]; AR.colIndex = mode[4]; }
    if (ch == 1) {
        // four pixels per lane: tile, canvas and validity bytes move as (unaligned) dwords
        const int x = (int)(bx * 256 + threadIdx.x) * 4;
        if (x >= tw) return;
        const size_t co = (size_t)cy * ccols + x0 + x;
        const uint8_t *tp = tile + (size_t)y * tw + x;
        const int nk = min(4, tw - x);
        uint32_t tb, pb = 0, mb = 0;
        if (nk == 4) { tb = *(const u32u1 *)tp; if (row_in) { pb = *(const u32u1 *)(pix + co); mb = *(const u32u1 *)(mask + co); } }
        else { tb = 0; for (int k = 0; k < nk; k++) { tb |= (uint32_t)tp[k] << (8 * k); if (row_in) { pb |= (uint32_t)pix[co + k] << (8 * k); mb |= (uint32_t)mask[co + k] << (8 * k); } } }
        uint32_t ob = tb;
        if (row_in) {
            const float war = analytic == 3 ? 1.f : analytic ? AR.a_row(i) : wAr[i], wbr = analytic == 3 ? AR.cb_row(i) : analytic ? AR.b_row(i) : wBr[i];
            ob = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = x0 + x + k - rx0;
                const int b = (tb >> (8 * k)) & 0xff;
                uint32_t o = (uint32_t)b;
                if (k < nk && j >= 0 && j < c) {
                    if (TG.on) o = trig_px(TG, corner, i, j, wbr, analytic == 3 ? AR.cb_col(j) : analytic ? 1.f : wBc[j], ((mb >> (8 * k)) & 0xff) != 0, (pb >> (8 * k)) & 0xff, b);
                    else {
                        float wA, wB;
                        if (corner) { wB = wbr * (analytic == 3 ? AR.cb_col(j) : wBc[j]); wA = 1 - wB; }
                        else if (analytic) { wA = war * AR.a_col(j); wB = wbr * AR.b_col(j); }
                        else { wA = war * wAc[j]; wB = wbr * wBc[j]; }
                        o = fade_px(wA, wB, ((mb >> (8 * k)) & 0xff) != 0, (pb >> (8 * k)) & 0xff, b);
                    }
                }
                ob |= o << (8 * k);
            }
        }
        if (nk == 4) { *(u32u1 *)(pix + co) = ob; *(u32u1 *)(mask + co) = 0x01010101u; }
        else for (int k = 0; k < nk; k++) { pix[co + k] = (uint8_t)(ob >> (8 * k)); mask[co + k] = 1; }
        return;
    }
    const int x = (int)(bx * 256 + threadIdx.x);
    if (x >= tw) return;
    const int cx = x0 + x;
    const size_t co = (size_t)cy * ccols + cx;
    const int j = cx - rx0;
    const bool in_roi = (row_in && j >= 0 && j < c);
    if (in_roi) {
        float wA, wB;
        const float cbr = analytic == 3 ? AR.cb_row(i) : analytic ? 1.f : wBr[i], cbc = analytic == 3 ? AR.cb_col(j) : analytic ? 1.f : wBc[j];
        if (corner) { wB = cbr * cbc; wA = 1 - wB; }
        else if (analytic) { wA = AR.a_row(i) * AR.a_col(j); wB = AR.b_row(i) * AR.b_col(j); }
        else { wA = wAr[i] * wAc[j]; wB = wBr[i] * wBc[j]; }
        const bool av = mask[co] != 0;
        for (int k = 0; k < ch; k++)
            pix[co * ch + k] = TG.on ? trig_px(TG, corner, i, j, cbr, cbc, av, (int)pix[co * ch + k], tile[((size_t)y * tw + x) * ch + k])
                                     : fade_px(wA, wB, av, (int)pix[co * ch + k], tile[((size_t)y * tw + x) * ch + k]);
    } else {
        for (int k = 0; k < ch; k++) pix[co * ch + k] = tile[((size_t)y * tw + x) * ch + k];
    }
    mask[co] = 1;
}

__global__ __launch_bounds__(256) void k_fuse_apply(uint8_t *pix, uint8_t *mask, int ccols, int ch,
                                                    const uint8_t *tile, int th, int tw, int y0, int x0,
                                                    int ry0, int rx0, int r, int c, const int *mode,
                                                    const float *wAr, const float *wAc, const float *wBr, const float *wBc, TrigGeom TG, int analytic)
{
    fuse_apply_block(pix, mask, ccols, ch, tile, th, tw, y0, x0, ry0, rx0, r, c, mode, wAr, wAc, wBr, wBc, TG, blockIdx.x, blockIdx.y, analytic);
}

__device__ __forceinline__ void paste_block(uint8_t *pix, uint8_t *mask, int ccols, int ch,
                                            const uint8_t *tile, int th, int tw, int y0, int x0, unsigned bx, unsigned by)
{
    const int x = (int)(bx * 256 + threadIdx.x);
    const int y = (int)by;
    if (x >= tw) return;
    const size_t co = (size_t)(y0 + y) * ccols + (x0 + x);
    for (int k = 0; k < ch; k++) pix[co * ch + k] = tile[((size_t)y * tw + x) * ch + k];
    mask[co] = 1;
}
__global__ __launch_bounds__(256) void k_paste(uint8_t *pix, uint8_t *mask, int ccols, int ch,
                                               const uint8_t *tile, int th, int tw, int y0, int x0)
{
    paste_block(pix, mask, ccols, ch, tile, th, tw, y0, x0, blockIdx.x, blockIdx.y);
}

// ---- int64 compatibility path (the reference's own array representation) -----------------------------
__global__ __launch_bounds__(256) void k_i64_stats_rows(I64View V, int r, int c, FuseStats *st, int *rowFirst, int *rowLast)
{
    const int i = blockIdx.x;
    __shared__ int sf[256], sl[256];
    __shared__ unsigned sv[256], s0[256], s1[256];
    int first = 0x7fffffff, last = -1; unsigned valid = 0, qlo = 0, qhi = 0;
    const int c2 = c / 2;
    for (int j = threadIdx.x; j < c; j += 256) {
        if (V.a_valid(i, j)) { first = min(first, j); last = max(last, j); }
        for (int k = 0; k < V.ch; k++) {
            long long a = V.a_raw(i, j, k);
            valid += a > -1;
            if (a > 0) { if (j < c2) qlo++; else qhi++; }
        }
    }
    sf[threadIdx.x] = first; sl[threadIdx.x] = last; sv[threadIdx.x] = valid; s0[threadIdx.x] = qlo; s1[threadIdx.x] = qhi;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            sf[threadIdx.x] = min(sf[threadIdx.x], sf[threadIdx.x + d]);
            sl[threadIdx.x] = max(sl[threadIdx.x], sl[threadIdx.x + d]);
            sv[threadIdx.x] += sv[threadIdx.x + d]; s0[threadIdx.x] += s0[threadIdx.x + d]; s1[threadIdx.x] += s1[threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        rowFirst[i] = sf[0] == 0x7fffffff ? -1 : sf[0];
        rowLast[i] = sl[0];
        atomicAdd(&st->valid, (unsigned long long)sv[0]);
        const bool top = i < r / 2;
        atomicAdd(&st->quad[top ? 0 : 1], (unsigned long long)s0[0]);
        atomicAdd(&st->quad[top ? 3 : 2], (unsigned long long)s1[0]);
    }
}
__global__ __launch_bounds__(256) void k_i64_stats_cols(I64View V, int r, int c, int *colFirst, int *colLast)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= c) return;
    int first = -1, last = -1;
    for (int i = 0; i < r; i++)
        if (V.a_valid(i, j)) { if (first < 0) first = i; last = i; }
    colFirst[j] = first; colLast[j] = last;
}
__global__ __launch_bounds__(256) void k_i64_apply(I64View V, int r, int c, const int *mode, const float *wAr, const float *wAc,
                                                   const float *wBr, const float *wBc, uint8_t *out, TrigGeom TG)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= c) return;
    float wA, wB;
    if (mode[0]) { wB = wBr[i] * wBc[j]; wA = 1 - wB; }
    else { wA = wAr[i] * wAc[j]; wB = wBr[i] * wBc[j]; }
    for (int k = 0; k < V.ch; k++) {
        long long a = V.a_raw(i, j, k), b = V.b_val(i, j, k);
        if (TG.on) {                                   // (u8-valued inputs: the reference's regions hold -1 or grey levels)
            out[((size_t)i * c + j) * V.ch + k] = trig_px(TG, mode[0] != 0, i, j, wBr[i], wBc[j], a >= 0, (int)a, (int)b);
            continue;
        }
        if (a < 0) a = b;
        double res = (double)wA * (double)a + (double)wB * (double)b;
        res = res < 0 ? 0 : res;
        res = res > 255 ? 255 : res;
        out[((size_t)i * c + j) * V.ch + k] = (uint8_t)res;
    }
}

// ---- host: mode decision + separable ramps (ImageFusion.py:204-239 and :43-190) -----------------------
// ---------------------------------------------------------------------------------------------------
// Weight ramps on the device (one workgroup).  fuseByFadeInAndFadeOut's strip ramps (ImageFusion.py:213-235) and
// getWeightsMatrix's corner ramps (ImageFusion.py:43-190) are separable; the Python scans for the first non-empty pixel
// become a first-match reduction over the per-row / per-column first/last-valid arrays, the ramp loops their closed
// forms (each index is written by exactly one loop iteration that survives, see the notes at the loops), with the
// reference's conventions kept: float32 strip arithmetic, double quotient cast to float32 in the corner ramps, index 0
// patched to 1, Python's negative-index wrap, and the geometries where the reference itself raises reported in out[5].
//   out: [0] corner mode, [1..4] info (mode, quadrant, rowIndex, colIndex), [5] error
// ---------------------------------------------------------------------------------------------------
// A record another workgroup of THIS launch produced (agent-scope atomic, or a write-through `sc1` store) is read with an agent-scope
// relaxed load (`global_load ... sc1`: bypasses this CU's L1) -- what lets k_fuse_stats_weights hand its statistics to the last workgroup
// without the two agent fences it had (round 6; MI355X_MICROARCH.md, "valid forms": sc1 stores + drained vmcnt + flag on the producing
// side, sc1 loads on the consuming side).  The standalone ramp kernel (k_fuse_weights, its inputs come from earlier launches) reads the
// same way; it costs it nothing.
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int pywrap(int i, int n) { return i < 0 ? i + n : i; }

// (first_encoded: the canvas path stores "first valid" positions as n - 1 - first so that every record starts at -1 and is a maximum;
//  bit 0: columns' first valid row, bit 1: rows' first valid column)
__device__ __forceinline__ void fuse_weights_body(int r, int c, int ch, int dx, int dy, int force_corner, const FuseStats *st,
                                                  const int *rowFirstRaw, const int *rowLastRaw, const int *colFirstRaw, const int *colLastRaw,
                                                  float *wAr, float *wAc, float *wBr, float *wBc, int *out, int *sticky_err, int first_encoded)
{
    struct First { const int *p; int n, enc; __device__ int operator[](int j) const { const int v = ld_agent(p + j); return (enc && v >= 0) ? n - 1 - v : v; } };
    struct Last { const int *p; __device__ int operator[](int j) const { return ld_agent(p + j); } };
    const Last colLast = {colLastRaw}, rowLast = {rowLastRaw};
    const First colFirst = {colFirstRaw, r, first_encoded & 1};
    const First rowFirst = {rowFirstRaw, c, (first_encoded >> 1) & 1};
    const int t = threadIdx.x;
    __shared__ int s_first, s_geom[4];          // first scan position with a non-zero candidate; index,rowIndex,colIndex,err
    for (int i = t; i < r; i += 256) { wAr[i] = 1.f; wBr[i] = 1.f; }
    for (int j = t; j < c; j += 256) { wAc[j] = 1.f; wBc[j] = 1.f; }
    if (t == 0) s_first = 0x7fffffff;
    __syncthreads();
    const double nel = (double)r * c * ch;
    const bool strip = !force_corner && (double)st->valid / nel > 0.65;
    if (strip) {
        if (c <= r) {                       // side-by-side strip: ramps along columns (float32 arithmetic)
            for (int i = t; i < c; i += 256) {
                const float f = (dy >= 0) ? (float)i : (float)(c - i);
                wAc[c - i - 1] = ((1.f * f) * 1.0f) / (float)c;
                wBc[i] = ((1.f * f) * 1.0f) / (float)c;
            }
        } else {                            // stacked strip: ramps along rows
            for (int i = t; i < r; i += 256) {
                const float f = (dx <= 0) ? (float)i : (float)(r - i);
                wAr[i] = ((1.f * f) * 1.0f) / (float)r;
                wBr[r - i - 1] = ((1.f * f) * 1.0f) / (float)r;
            }
        }
        if (t == 0) { out[0] = 0; out[1] = 0; out[2] = -1; out[3] = 0; out[4] = 0; out[5] = 0; }
        return;
    }
    int index = 0;
    for (int q = 1; q < 4; q++) if (st->quad[q] < st->quad[index]) index = q;
    // the Python loop walks the columns (from the right for index 2/3, from the left otherwise) until rowIndex becomes non-zero
    const bool from_right = index == 2 || index == 3;
    const int jlo = from_right ? 1 : 0;
    int mine = 0x7fffffff;
    for (int j = jlo + t; j < c; j += 256) {
        const int col = from_right ? c - j : j;
        int cand = 0;
        if (index == 2 || index == 1) { if (colLast[col] >= 0) cand = colLast[col] + 1; }
        else                          { if (colFirst[col] >= 0) cand = colFirst[col] - 1; }
        if (cand != 0) { mine = j; break; }
    }
    atomicMin(&s_first, mine);
    __syncthreads();
    if (t == 0) {
        int rowIndex = 0, colIndex = 0, err = 0;
        if (s_first != 0x7fffffff) {
            const int col = from_right ? c - s_first : s_first;
            rowIndex = (index == 2 || index == 1) ? colLast[col] + 1 : colFirst[col] - 1;
        }
        if (rowIndex >= r) err = 1;
        else {
            const int rr = pywrap(rowIndex, r);
            if (from_right) { if (rowLast[rr] >= 0) colIndex = rowLast[rr] + 1; }
            else            { if (rowFirst[rr] >= 0) colIndex = rowFirst[rr] - 1; }
        }
        s_geom[0] = index; s_geom[1] = rowIndex; s_geom[2] = colIndex; s_geom[3] = err;
    }
    __syncthreads();
    const int rowIndex = s_geom[1], colIndex = s_geom[2];
    int err = s_geom[3];
    if (!err) {
        // rows.  index 2 / 1: for i in range(rowIndex + 1): wB[ri - i] = (ri - i) / ri  (ri = rowIndex, 0 patched to 1): indices ri..ri-rowIndex, all distinct
        //        index 3 / 0: for i in range(rowIndex, row): wB[i] = (row - i - 1) / (row - ri - 1); a negative start only writes wrapped indices
        //                     that later iterations overwrite, so the surviving writes are i = max(rowIndex, 0) .. row - 1
        if (index == 2 || index == 1) {
            const int n = rowIndex + 1, ri = rowIndex == 0 ? 1 : rowIndex;
            for (int i = t; i < n; i += 256) {
                const int idx = ri - i;
                if (idx >= r) { err = 1; break; }
                wBr[pywrap(idx, r)] = (float)((double)(ri - i) * 1 / ri);
            }
        } else {
            const int ri = rowIndex == 0 ? 1 : rowIndex;
            if (rowIndex < r && r - ri - 1 == 0) err = 1;
            else for (int i = max(rowIndex, 0) + t; i < r; i += 256) wBr[i] = (float)((double)(r - i - 1) * 1 / (r - ri - 1));
        }
        if (index == 2 || index == 3) {
            const int n = colIndex + 1, ci = colIndex == 0 ? 1 : colIndex;
            for (int i = t; i < n; i += 256) {
                const int idx = ci - i;
                if (idx >= c) { err = 1; break; }
                wBc[pywrap(idx, c)] = (float)((double)(ci - i) * 1 / ci);
            }
        } else {
            const int ci = colIndex == 0 ? 1 : colIndex;
            if (colIndex < c && c - ci - 1 == 0) err = 1;
            else for (int i = max(colIndex, 0) + t; i < c; i += 256) wBc[i] = (float)((double)(c - i - 1) * 1 / (c - ci - 1));
        }
    }
    if (err) { atomicOr(&out[5], 1); if (sticky_err) atomicOr(sticky_err, 1); }
    if (t == 0) { out[0] = 1; out[1] = 1; out[2] = s_geom[0]; out[3] = rowIndex; out[4] = colIndex; }
}

__global__ __launch_bounds__(256) void k_fuse_weights(int r, int c, int ch, int dx, int dy, int force_corner, const FuseStats *st,
                                                      const int *rowFirst, const int *rowLast, const int *colFirstRaw, const int *colLast,
                                                      float *wAr, float *wAc, float *wBr, float *wBc, int *out, int *sticky_err, int first_encoded)
{
    fuse_weights_body(r, c, ch, dx, dy, force_corner, st, rowFirst, rowLast, colFirstRaw, colLast, wAr, wAc, wBr, wBc, out, sticky_err, first_encoded);
}

// ---------------------------------------------------------------------------------------------------
// ONE statistics launch per fused tile (canvas path).  A wave owns FUSE_SB rows x 256 columns of the ROI, the four waves of a workgroup
// sit 4 x 1, 2 x 2 or 1 x 4 (tall overlap strips) over it: a lane walks its four columns down the band with the dwords of all FUSE_SB
// rows in flight (first / last valid row per column in registers, combined over the workgroup's sub-bands in LDS, one atomicMax each),
// every row's first / last valid column comes from two wave ballots (one atomicMax per wave and row), the occupancy and quadrant counts
// from one block reduction into the workgroup's own slot.  The workgroup that finishes LAST (a ticket counter) builds the ramps right there -- no separate launch -- and then
// puts every record back to its initial value, so the scratch (allocated with the canvas, sized by its rows + cols) needs no memset per
// tile either: a fused tile is this launch + the blend.  (Before: two memsets + rows, columns, ramps, blend = six dependent launches;
// the chain of a tile is still latency-bound at ~50 us -- see DESIGN.md for what was tried.)
// ---------------------------------------------------------------------------------------------------
#define FUSE_SB 16
#define FUSE_NW 4                    // waves of a statistics workgroup (16-wave workgroups, i.e. a quarter of the tickets: 8 % slower)
struct FuseCanvasScratch { unsigned *slots; int *out; unsigned *done; int *rowFirstEnc, *rowLast, *colFirstEnc, *colLast; float *wAr, *wBr, *wAc, *wBc; };

template <int FUSE_SBT, int FUSE_UR, bool GEOM>
__device__ __forceinline__ void fuse_stats_block(const CanvasView &V, int r, int c, const FuseCanvasScratch &S, int wx_n, unsigned bx, unsigned by, unsigned nbx)
{
    __shared__ unsigned s_cnt[5][FUSE_NW];
    __shared__ int s_col[2][FUSE_NW * 256];                                   // [first | last][wave row][column of the workgroup]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // The four waves are laid out wx_n wide x 4 / wx_n high (1 x 4 for ROIs of up to 256 columns -- the tall overlap strips of a mosaic row --
    // so that no wave idles beside the ROI and a column receives one atomic per 4 sub-bands).
    const int wx = wave % wx_n, wy = wave / wx_n, wy_n = FUSE_NW / wx_n, wcols = wx_n * 256;
    const int jl = (wx * 64 + lane) * 4;                                // column inside the workgroup
    const int j = (int)bx * wx_n * 256 + jl;                         // four adjacent columns per lane: validity bytes as one dword
    const int i0 = min(((int)by * wy_n + wy) * FUSE_SBT, r), i1 = min(i0 + FUSE_SBT, r);
    const int c2 = c / 2, r2 = r / 2;
    const int nk = max(0, min(4, c - j));
    int first[4] = {-1, -1, -1, -1}, last[4] = {-1, -1, -1, -1};
    unsigned valid = 0, q_tl = 0, q_bl = 0, q_br = 0, q_tr = 0;
    // The rows of a band are independent, but the atomics keep the compiler from hoisting the loads of row i + 1 above the work of row i,
    // which left 32 dependent memory round trips per lane.  Validity (and, for gray canvases, pixel) dwords of FUSE_UR rows are fetched
    // up front, then consumed.
    for (int ib = i0; ib < i1; ib += FUSE_UR) {
        uint32_t mm[FUSE_UR], pp[FUSE_UR];
#pragma unroll
        for (int u = 0; u < FUSE_UR; u++) {
            mm[u] = 0; pp[u] = 0;
            const int i = ib + u;
            if (i < i1 && nk > 0) {
                const size_t o = (size_t)(V.ry0 + i) * V.ccols + V.rx0 + j;
                if (nk == 4) { mm[u] = *(const u32u1 *)(V.mask + o); if (V.ch == 1) pp[u] = *(const u32u1 *)(V.pix + o); }
                else for (int k = 0; k < nk; k++) { mm[u] |= (uint32_t)V.mask[o + k] << (8 * k); if (V.ch == 1) pp[u] |= (uint32_t)V.pix[o + k] << (8 * k); }
            }
        }
#pragma unroll
        for (int u = 0; u < FUSE_UR; u++) {
            const int i = ib + u;
            if (i >= i1) continue;
            const uint32_t m = mm[u], pz = pp[u];
            unsigned pos_lo = 0, pos_hi = 0;
            if (m) {
                if (V.ch == 1) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if ((m >> (8 * k)) & 0xff) {
                            valid += 1;
                            const unsigned pos = ((pz >> (8 * k)) & 0xff) != 0;
                            if (j + k < c2) pos_lo += pos; else pos_hi += pos;
                        }
                } else {
                    const size_t o = (size_t)(V.ry0 + i) * V.ccols + V.rx0 + j;
                    for (int k = 0; k < nk; k++)
                        if ((m >> (8 * k)) & 0xff) {
                            valid += V.ch;
                            unsigned pos = 0;
                            for (int q = 0; q < V.ch; q++) pos += V.pix[(o + k) * V.ch + q] > 0;
                            if (j + k < c2) pos_lo += pos; else pos_hi += pos;
                        }
                }
                if (GEOM) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if ((m >> (8 * k)) & 0xff) { if (first[k] < 0) first[k] = i; last[k] = i; }
                }
            }
            if (i < r2) { q_tl += pos_lo; q_tr += pos_hi; } else { q_bl += pos_lo; q_br += pos_hi; }
            // first / last valid column of row i inside this wave's 256 columns (GEOM = false: the host derives the geometry from the canvas's
            // rectangle list, the blocks only count)
            const unsigned long long any = GEOM ? __ballot(m != 0) : 0ull;
            if (any) {
                const int lo_lane = __ffsll((long long)any) - 1, hi_lane = 63 - __clzll((long long)any);
                if (lane == lo_lane) atomicMax(&S.rowFirstEnc[i], c - 1 - (j + (__ffs((int)m) - 1) / 8));
                if (lane == hi_lane) atomicMax(&S.rowLast[i], j + (31 - __clz((int)m)) / 8);
            }
        }
    }
    if (!GEOM) {
    } else if (wy_n == 1) {
        for (int k = 0; k < nk; k++)
            if (last[k] >= 0) { atomicMax(&S.colLast[j + k], last[k]); atomicMax(&S.colFirstEnc[j + k], r - 1 - first[k]); }
    } else {
        // sub-bands are in row order: the first valid row of a column is that of the lowest wy with one, the last valid row that of the highest
#pragma unroll
        for (int k = 0; k < 4; k++) { s_col[0][wy * wcols + jl + k] = first[k]; s_col[1][wy * wcols + jl + k] = last[k]; }
        __syncthreads();
        if (wy == 0) {
            for (int k = 0; k < nk; k++) {
                int f = -1, l = -1;
                for (int y = 0; y < wy_n; y++) {
                    const int fy = s_col[0][y * wcols + jl + k], ly = s_col[1][y * wcols + jl + k];
                    if (f < 0) f = fy;
                    if (ly >= 0) l = ly;
                }
                if (l >= 0) { atomicMax(&S.colLast[j + k], l); atomicMax(&S.colFirstEnc[j + k], r - 1 - f); }
            }
        }
    }
    // block reduction of the five counts: wave shuffles, then one set of atomics per workgroup
    unsigned v5[5] = {valid, q_tl, q_bl, q_br, q_tr};
#pragma unroll
    for (int q = 0; q < 5; q++) {
        unsigned x = v5[q];
        for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
        if (lane == 0) s_cnt[q][wave] = x;
    }
    __syncthreads();
    // the five counts go to the workgroup's own slot (plain stores: five atomics per workgroup on ONE cache line were what the launch waited for)
    const unsigned wg = by * nbx + bx;
    if (threadIdx.x < 5)
    {
        unsigned x = 0;
        for (int w = 0; w < FUSE_NW; w++) x += s_cnt[threadIdx.x][w];
        __hip_atomic_store(&S.slots[(size_t)wg * 8 + threadIdx.x], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through (sc1): visible to the last workgroup without a fence
    }
}

// one workgroup, after every statistics block of the tile is visible: sums the slots, builds the ramps, puts the records back to -1
__device__ __forceinline__ void fuse_ramps_tail(int r, int c, int ch, int dx, int dy, const FuseCanvasScratch &S, unsigned nwg, int *sticky_err)
{
    __shared__ FuseStats s_st;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 8) S.out[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_st.valid = 0; s_st.quad[0] = 0; s_st.quad[1] = 0; s_st.quad[2] = 0; s_st.quad[3] = 0; }
    __syncthreads();
    {
        unsigned long long sum5[5] = {0, 0, 0, 0, 0};
        for (unsigned w = threadIdx.x; w < nwg; w += 256)
#pragma unroll
            for (int q = 0; q < 5; q++) sum5[q] += ld_agent(&S.slots[(size_t)w * 8 + q]);
#pragma unroll
        for (int q = 0; q < 5; q++) {
            unsigned long long x = sum5[q];
            for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
            if (lane == 0 && x) atomicAdd(q == 0 ? &s_st.valid : &s_st.quad[q - 1], x);
        }
    }
    __syncthreads();
    fuse_weights_body(r, c, ch, dx, dy, 0, &s_st, S.rowFirstEnc, S.rowLast, S.colFirstEnc, S.colLast, S.wAr, S.wAc, S.wBr, S.wBc, S.out, sticky_err, 3);
    __syncthreads();
    // records back to their initial values for the next tile
    for (int i = threadIdx.x; i < r; i += 256) { S.rowFirstEnc[i] = -1; S.rowLast[i] = -1; }
    for (int jj = threadIdx.x; jj < c; jj += 256) { S.colFirstEnc[jj] = -1; S.colLast[jj] = -1; }
    if (threadIdx.x == 0) *S.done = 0;
}

template <int FUSE_SBT, int FUSE_UR>
__global__ __launch_bounds__(FUSE_NW * 64) void k_fuse_stats_weights(CanvasView V, int r, int c, FuseCanvasScratch S, int dx, int dy, int *sticky_err, int wx_n)
{
    __shared__ int s_last;
    fuse_stats_block<FUSE_SBT, FUSE_UR, true>(V, r, c, S, wx_n, blockIdx.x, blockIdx.y, gridDim.x);
    // The last workgroup to arrive builds the ramps.  Round 6: no agent fences (they were 20 % of a mosaic: ~11 us per tile,
    // profiles/r06_ab_fuse_fences.txt).  Everything a statistics block publishes is either an agent-scope atomic (row / column records) or a
    // write-through store (its slot); a thread's are complete when its vmcnt has drained, the barrier collects the workgroup's, and only
    // then does thread 0 take the ticket -- so whoever draws the last ticket finds every block's records at the agent's coherence point and
    // reads them with sc1 loads (ld_agent); no line of them is in this CU's L1 or this XCD's L2 (they are only ever touched by atomics and
    // sc1 accesses inside this launch, and the ramps / resets the tail writes are for the NEXT launch: kernel boundary).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(S.done, 1u) == gridDim.x * gridDim.y - 1;
    __syncthreads();
    if (!s_last || threadIdx.x >= 256) return;          // the ramps are a 256-thread job (finished waves do not count at the barriers)
    fuse_ramps_tail(r, c, V.ch, dx, dy, S, gridDim.x * gridDim.y, sticky_err);
}

// A corner ROI on a canvas whose validity the host knows (the rectangle list): getWeightsMatrix's geometry -- rowIndex, colIndex, the degenerate
// cases -- is a function of the VALIDITY pattern and of `index`, the quadrant with the fewest non-zero elements; only that count needs the
// pixels.  The host evaluates the geometry for all four possible quadrants (CornerPick), the blocks count, the last workgroup sums the slots,
// picks the quadrant and publishes {1, 1, index, rowIndex, colIndex, err}: no row / column records, no atomics per row, no dependent record
// loads in the tail, no ramp arrays (k_fuse_apply forms the corner ramps from the three numbers).
struct CornerPick { int rowIndex[4], colIndex[4], err[4]; };

template <int FUSE_SBT, int FUSE_UR>
__global__ __launch_bounds__(FUSE_NW * 64) void k_fuse_counts_pick(CanvasView V, int r, int c, FuseCanvasScratch S, CornerPick P, int *sticky_err, int wx_n)
{
    __shared__ int s_last;
    fuse_stats_block<FUSE_SBT, FUSE_UR, false>(V, r, c, S, wx_n, blockIdx.x, blockIdx.y, gridDim.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(S.done, 1u) == gridDim.x * gridDim.y - 1;
    __syncthreads();
    if (!s_last || threadIdx.x >= 64) return;
    const unsigned nwg = gridDim.x * gridDim.y;
    unsigned long long q4[4] = {0, 0, 0, 0};
    for (unsigned w = threadIdx.x; w < nwg; w += 64)
#pragma unroll
        for (int q = 0; q < 4; q++) q4[q] += ld_agent(&S.slots[(size_t)w * 8 + 1 + q]);
#pragma unroll
    for (int q = 0; q < 4; q++)
        for (int d = 32; d > 0; d >>= 1) q4[q] += __shfl_down(q4[q], d, 64);
    if (threadIdx.x == 0) {
        int index = 0;
        for (int q = 1; q < 4; q++) if (q4[q] < q4[index]) index = q;
        S.out[0] = 1; S.out[1] = 1; S.out[2] = index; S.out[3] = P.rowIndex[index]; S.out[4] = P.colIndex[index]; S.out[5] = P.err[index];
        if (P.err[index] && sticky_err) atomicOr(sticky_err, 1);
        *S.done = 0;
    }
}

struct FuseScratch { FuseStats *st; int *rowFirst, *rowLast, *colFirst, *colLast; float *wAr, *wAc, *wBr, *wBc; int *out; };

static int fuse_scratch(vfsms_ctx *ctx, int r, int c, FuseScratch *S)
{
    S->st = (FuseStats *)ctx_arena_alloc(ctx, sizeof(FuseStats));
    S->rowFirst = (int *)ctx_arena_alloc(ctx, sizeof(int) * r); S->rowLast = (int *)ctx_arena_alloc(ctx, sizeof(int) * r);
    S->colFirst = (int *)ctx_arena_alloc(ctx, sizeof(int) * c); S->colLast = (int *)ctx_arena_alloc(ctx, sizeof(int) * c);
    S->wAr = (float *)ctx_arena_alloc(ctx, sizeof(float) * r); S->wBr = (float *)ctx_arena_alloc(ctx, sizeof(float) * r);
    S->wAc = (float *)ctx_arena_alloc(ctx, sizeof(float) * c); S->wBc = (float *)ctx_arena_alloc(ctx, sizeof(float) * c);
    S->out = (int *)ctx_arena_alloc(ctx, sizeof(int) * 8);
    if (!S->out) { vfsms_set_error("arena exhausted in fuse"); return VFSMS_ERR_CAPACITY; }
    return VFSMS_OK;
}

// launch the ramp kernel; `finish_weights` later brings back its 6 status ints (and the ramps when a caller wants them) in one sync
static int launch_weights(vfsms_ctx *ctx, const FuseScratch &S, int r, int c, int ch, int dx, int dy, int force_corner = 0, int *sticky_err = nullptr,
                          int first_encoded = 0, bool out_cleared = false)
{
    if (!out_cleared) HIP_TRY(hipMemsetAsync(S.out, 0, sizeof(int) * 8, ctx->stream));
    hipLaunchKernelGGL(k_fuse_weights, dim3(1), dim3(256), 0, ctx->stream, r, c, ch, dx, dy, force_corner, S.st, S.rowFirst, S.rowLast,
                       S.colFirst, S.colLast, S.wAr, S.wAc, S.wBr, S.wBc, S.out, sticky_err, first_encoded);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

static int finish_weights(vfsms_ctx *ctx, const FuseScratch &S, int r, int c, int32_t *info, float *h_ramps = nullptr)
{
    int out[8];
    HIP_TRY(hipMemcpyAsync(out, S.out, sizeof(out), hipMemcpyDeviceToHost, ctx->stream));
    if (h_ramps) {                                 // [wA_r | wB_r | wA_c | wB_c]
        HIP_TRY(hipMemcpyAsync(h_ramps, S.wAr, sizeof(float) * r, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(h_ramps + r, S.wBr, sizeof(float) * r, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(h_ramps + 2 * r, S.wAc, sizeof(float) * c, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(h_ramps + 2 * r + c, S.wBc, sizeof(float) * c, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (info) for (int k = 0; k < 4; k++) info[k] = out[1 + k];
    if (out[5]) {
        vfsms_set_error("fuse: degenerate corner geometry (the reference's getWeightsMatrix raises here)");
        return VFSMS_ERR_BAD_ARG;
    }
    return VFSMS_OK;
}

// average / maximum / minimum (ImageFusion.py:12-41) behind fuseImage's pre-processing (Stitcher.py:498-504): empty (-1) and zero
// elements of A are filled from B, then zero elements of B from the updated A, element by element; outside the ROI the tile is
// pasted.  mode: 0 average = uint8((A + B) / 2), 1 maximum, 2 minimum.
__global__ __launch_bounds__(256) void k_fuse_simple(uint8_t *pix, uint8_t *mask, int ccols, int ch, const uint8_t *tile, int h, int w,
                                                     int y0, int x0, int ry0, int rx0, int r, int c, int mode)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= w || i >= h) return;
    const int cy = y0 + i, cx = x0 + j;
    const size_t idx = (size_t)cy * ccols + cx;
    const bool inroi = cy >= ry0 && cy < ry0 + r && cx >= rx0 && cx < rx0 + c;
    const int m = mask[idx];
    for (int k = 0; k < ch; k++) {
        const int B0 = tile[((size_t)i * w + j) * ch + k];
        int res = B0;
        if (inroi) {
            const int A0 = m ? pix[idx * ch + k] : 0;
            const int A1 = A0 == 0 ? B0 : A0;
            const int B1 = B0 == 0 ? A1 : B0;
            res = mode == 0 ? (A1 + B1) >> 1 : mode == 1 ? max(A1, B1) : min(A1, B1);
        }
        pix[idx * ch + k] = (uint8_t)res;
    }
    mask[idx] = 1;
}

// ---- the canvas's validity, known to the host ------------------------------------------------------------------------------------------------
// Every write to a canvas is a whole tile rectangle (k_paste, k_fuse_apply and k_fuse_simple mark all of it valid), so the validity mask IS the
// union of the rectangles placed so far.  fuseByFadeInAndFadeOut needs the COUNT of valid elements of the ROI to choose between its strip and
// its corner branch (ImageFusion.py:201: count / size > 0.65); the strip branch then needs nothing else from the canvas -- its ramps are closed
// forms of (row, col, dx, dy).  The host therefore counts the valid pixels of the ROI from the rectangle list (exact: coordinate compression
// over the few rectangles that meet the ROI), and a strip tile is ONE launch (k_fuse_apply with analytic ramps) instead of statistics + blend.
static void canvas_mark(CanvasRec *cv, int y0, int x0, int h, int w)
{
    const int32_t q[4] = {std::max(y0, 0), std::max(x0, 0), std::min(y0 + h, cv->rows), std::min(x0 + w, cv->cols)};
    if (q[2] > q[0] && q[3] > q[1]) cv->placed.insert(cv->placed.end(), q, q + 4);
}

static long long canvas_valid_area(const CanvasRec *cv, int ry0, int rx0, int ry1, int rx1)
{
    std::vector<int> ys, xs, hit;
    for (size_t k = 0; k + 3 < cv->placed.size(); k += 4) {
        const int a0 = std::max(cv->placed[k], ry0), b0 = std::max(cv->placed[k + 1], rx0), a1 = std::min(cv->placed[k + 2], ry1), b1 = std::min(cv->placed[k + 3], rx1);
        if (a1 > a0 && b1 > b0) { hit.push_back(a0); hit.push_back(b0); hit.push_back(a1); hit.push_back(b1); ys.push_back(a0); ys.push_back(a1); xs.push_back(b0); xs.push_back(b1); }
    }
    std::sort(ys.begin(), ys.end()); ys.erase(std::unique(ys.begin(), ys.end()), ys.end());
    std::sort(xs.begin(), xs.end()); xs.erase(std::unique(xs.begin(), xs.end()), xs.end());
    long long area = 0;
    for (size_t a = 0; a + 1 < ys.size(); a++)
        for (size_t b = 0; b + 1 < xs.size(); b++) {
            bool in = false;
            for (size_t k = 0; k < hit.size() && !in; k += 4) in = hit[k] <= ys[a] && ys[a + 1] <= hit[k + 2] && hit[k + 1] <= xs[b] && xs[b + 1] <= hit[k + 3];
            if (in) area += (long long)(ys[a + 1] - ys[a]) * (xs[b + 1] - xs[b]);
        }
    return area;
}

int canvas_blend_device(vfsms_ctx *ctx, CanvasRec *cv, const uint8_t *d_tile, int h, int w, int y0, int x0,
                        int ry0, int rx0, int ry1, int rx1, int mode)
{
    const int r = ry1 - ry0 > 0 ? ry1 - ry0 : 0, c = rx1 - rx0 > 0 ? rx1 - rx0 : 0;
    ProfScope ps(ctx, "fuse");
    hipLaunchKernelGGL(k_fuse_simple, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, cv->pix, cv->mask, cv->cols, cv->ch,
                       d_tile, h, w, y0, x0, ry0, rx0, r, c, mode);
    HIP_TRY(hipGetLastError());
    canvas_mark(cv, y0, x0, h, w);
    return VFSMS_OK;
}

int canvas_paste_device(vfsms_ctx *ctx, CanvasRec *cv, const uint8_t *d_tile, int h, int w, int y0, int x0)
{
    hipLaunchKernelGGL(k_paste, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, cv->pix, cv->mask, cv->cols, cv->ch,
                       d_tile, h, w, y0, x0);
    HIP_TRY(hipGetLastError());
    canvas_mark(cv, y0, x0, h, w);
    return VFSMS_OK;
}

int canvas_fuse_device(vfsms_ctx *ctx, CanvasRec *cv, const uint8_t *d_tile, int h, int w, int y0, int x0,
                       int ry0, int rx0, int ry1, int rx1, int dx, int dy, int32_t *info, int method)
{
    const int r = ry1 - ry0, c = rx1 - rx0;
    if (r <= 0 || c <= 0) return canvas_paste_device(ctx, cv, d_tile, h, w, y0, x0);
    ProfScope ps(ctx, "fuse");
    const TrigGeom TG = {method == 1, r, c, dx, dy};
    const dim3 agrid(cv->ch == 1 ? (w + 1023) / 1024 : (w + 255) / 256, h);
    const char *env_an = getenv("VFSMS_FUSE_ANALYTIC");                      // 0: always run the statistics kernel (A/B runs, tests)
    const bool analytic_on = !(env_an && atoi(env_an) == 0);
    if (analytic_on) {
        // fuseByFadeInAndFadeOut's own test (ImageFusion.py:201), on the count the statistics kernel would have produced: valid elements = valid
        // pixels x channels
        const long long valid = canvas_valid_area(cv, ry0, rx0, ry1, rx1) * cv->ch;
        const double nel = (double)r * c * cv->ch;
        if ((double)valid / nel > 0.65) {
            hipLaunchKernelGGL(k_fuse_apply, agrid, dim3(256), 0, ctx->stream, cv->pix, cv->mask, cv->cols, cv->ch, d_tile, h, w, y0, x0, ry0, rx0, r, c,
                               (const int *)nullptr, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, TG, c <= r ? 1 : 2);
            HIP_TRY(hipGetLastError());
            canvas_mark(cv, y0, x0, h, w);
            if (info) { info[0] = 0; info[1] = -1; info[2] = 0; info[3] = 0; }      // what the ramp kernel reports for a strip: mode 0, no corner index
            return VFSMS_OK;
        }
    }
    // the canvas's own scratch (allocated and initialised once: canvas_scratch_bytes / canvas_scratch_init): st | out | done | rows | cols | ramps
    FuseCanvasScratch S;
    char *base = (char *)cv->scratch;
    S.out = (int *)(base + 64); S.done = (unsigned *)(base + 128);
    int *ib = (int *)(base + 256);
    S.rowFirstEnc = ib; S.rowLast = ib + cv->rows; S.colFirstEnc = ib + 2 * (size_t)cv->rows; S.colLast = S.colFirstEnc + cv->cols;
    float *fb = (float *)(S.colLast + cv->cols);
    S.wAr = fb; S.wBr = fb + cv->rows; S.wAc = fb + 2 * (size_t)cv->rows; S.wBc = S.wAc + cv->cols;
    S.slots = (unsigned *)(S.wBc + cv->cols);          // 8 dwords per statistics workgroup (canvas_scratch_bytes bounds their number)
    CanvasView V;
    V.pix = cv->pix; V.mask = cv->mask; V.ccols = cv->cols; V.ch = cv->ch; V.ry0 = ry0; V.rx0 = rx0;
    V.tile = d_tile; V.tw = w; V.ty0 = ry0 - y0; V.tx0 = rx0 - x0;
    const int wx_n = c <= 256 ? 1 : c <= 512 ? 2 : 4, wy_n = FUSE_NW / wx_n;
    const dim3 sgrid((c + 256 * wx_n - 1) / (256 * wx_n), (r + FUSE_SB * wy_n - 1) / (FUSE_SB * wy_n));
    if (analytic_on) {
        // a corner ROI: first / last valid column of every row and first / last valid row of every column from the rectangles that meet the ROI,
        // then getWeightsMatrix's scan (fuse_weights_body) for each of the four quadrants `index` could turn out to be
        std::vector<int> rec((size_t)2 * r + 2 * c, -1);
        int *rowF = rec.data(), *rowL = rowF + r, *colF = rowL + r, *colL = colF + c;
        for (size_t k = 0; k + 3 < cv->placed.size(); k += 4) {
            const int a0 = std::max(cv->placed[k], ry0) - ry0, b0 = std::max(cv->placed[k + 1], rx0) - rx0;
            const int a1 = std::min(cv->placed[k + 2], ry1) - ry0, b1 = std::min(cv->placed[k + 3], rx1) - rx0;
            if (a1 <= a0 || b1 <= b0) continue;
            for (int i = a0; i < a1; i++) { rowF[i] = rowF[i] < 0 ? b0 : std::min(rowF[i], b0); rowL[i] = std::max(rowL[i], b1 - 1); }
            for (int jj = b0; jj < b1; jj++) { colF[jj] = colF[jj] < 0 ? a0 : std::min(colF[jj], a0); colL[jj] = std::max(colL[jj], a1 - 1); }
        }
        CornerPick P;
        for (int index = 0; index < 4; index++) {
            const bool from_right = index == 2 || index == 3, by_last = index == 2 || index == 1;
            int rowIndex = 0, colIndex = 0, err = 0;
            for (int jj = from_right ? 1 : 0; jj < c; jj++) {
                const int col = from_right ? c - jj : jj;
                int cand = 0;
                if (by_last) { if (colL[col] >= 0) cand = colL[col] + 1; }
                else         { if (colF[col] >= 0) cand = colF[col] - 1; }
                if (cand != 0) { rowIndex = cand; break; }
            }
            if (rowIndex >= r) err = 1;
            else {
                const int rr = rowIndex < 0 ? rowIndex + r : rowIndex;
                if (from_right) { if (rowL[rr] >= 0) colIndex = rowL[rr] + 1; }
                else            { if (rowF[rr] >= 0) colIndex = rowF[rr] - 1; }
            }
            if (!err) {                                  // the degenerate cases of the ramp loops (fuse_weights_body)
                const int ri = rowIndex == 0 ? 1 : rowIndex, ci = colIndex == 0 ? 1 : colIndex;
                if (by_last) { if (ri >= r) err = 1; } else if (rowIndex < r && r - ri - 1 == 0) err = 1;
                if (from_right) { if (ci >= c) err = 1; } else if (colIndex < c && c - ci - 1 == 0) err = 1;
            }
            P.rowIndex[index] = rowIndex; P.colIndex[index] = colIndex; P.err[index] = err;
        }
        hipLaunchKernelGGL((k_fuse_counts_pick<FUSE_SB, FUSE_SB>), sgrid, dim3(FUSE_NW * 64), 0, ctx->stream, V, r, c, S, P, cv->d_err, wx_n);
        hipLaunchKernelGGL(k_fuse_apply, agrid, dim3(256), 0, ctx->stream, cv->pix, cv->mask, cv->cols, cv->ch, d_tile, h, w, y0, x0, ry0, rx0, r, c,
                           (const int *)S.out, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, TG, 3);
    } else {
        hipLaunchKernelGGL((k_fuse_stats_weights<FUSE_SB, FUSE_SB>), sgrid, dim3(FUSE_NW * 64), 0, ctx->stream, V, r, c, S, dx, dy, cv->d_err, wx_n);
        hipLaunchKernelGGL(k_fuse_apply, agrid, dim3(256), 0, ctx->stream, cv->pix, cv->mask, cv->cols, cv->ch,
                           d_tile, h, w, y0, x0, ry0, rx0, r, c, (const int *)S.out, (const float *)S.wAr, (const float *)S.wAc, (const float *)S.wBr, (const float *)S.wBc, TG, 0);
    }
    HIP_TRY(hipGetLastError());
    canvas_mark(cv, y0, x0, h, w);
    if (!info) return VFSMS_OK;          // no readback wanted: a degenerate geometry is latched in the canvas and reported by the download
    int out[8];
    HIP_TRY(hipMemcpyAsync(out, S.out, sizeof(out), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 4; k++) info[k] = out[1 + k];
    if (out[5]) {
        vfsms_set_error("fuse: degenerate corner geometry (the reference's getWeightsMatrix raises here)");
        return VFSMS_ERR_BAD_ARG;
    }
    return VFSMS_OK;
}

size_t canvas_scratch_bytes(int rows, int cols)
{
    const size_t slots = ((size_t)rows / FUSE_SB + 2) * ((size_t)cols / 1024 + 2);       // >= workgroups of any ROI inside the canvas, every wave layout
    return 256 + (sizeof(int) * 2 + sizeof(float) * 2) * ((size_t)rows + cols) + 32 * slots + 256;
}
int canvas_scratch_init(vfsms_ctx *ctx, CanvasRec *cv)
{
    HIP_TRY(hipMemsetAsync(cv->scratch, 0, 256, ctx->stream));
    HIP_TRY(hipMemsetAsync((char *)cv->scratch + 256, 0xff, sizeof(int) * 2 * ((size_t)cv->rows + cv->cols), ctx->stream));
    return VFSMS_OK;
}

// A, B: device int64 [r][c][ch]; out: device u8
int fuse_i64_device(vfsms_ctx *ctx, const long long *dA, const long long *dB, int r, int c, int ch, int dx, int dy,
                    uint8_t *d_out, int32_t *info, int method)
{
    FuseScratch S;
    TRY(fuse_scratch(ctx, r, c, &S));
    HIP_TRY(hipMemsetAsync(S.st, 0, sizeof(FuseStats), ctx->stream));
    I64View V; V.A = dA; V.B = dB; V.c = c; V.ch = ch;
    hipLaunchKernelGGL(k_i64_stats_rows, dim3(r), dim3(256), 0, ctx->stream, V, r, c, S.st, S.rowFirst, S.rowLast);
    hipLaunchKernelGGL(k_i64_stats_cols, dim3((c + 255) / 256), dim3(256), 0, ctx->stream, V, r, c, S.colFirst, S.colLast);
    TRY(launch_weights(ctx, S, r, c, ch, dx, dy));
    hipLaunchKernelGGL(k_i64_apply, dim3((c + 255) / 256, r), dim3(256), 0, ctx->stream, V, r, c, S.out,
                       S.wAr, S.wAc, S.wBr, S.wBc, d_out, TrigGeom{method == 1, r, c, dx, dy});
    HIP_TRY(hipGetLastError());
    return finish_weights(ctx, S, r, c, info);
}

// separable ramps only (ImageFusion.getWeightsMatrix when force_corner, else the mode fuseByFadeInAndFadeOut picks)
int fuse_i64_ramps(vfsms_ctx *ctx, const long long *dA, int r, int c, int ch, int dx, int dy, int force_corner,
                   float *h_ramps, int32_t *info)
{
    FuseScratch S;
    TRY(fuse_scratch(ctx, r, c, &S));
    HIP_TRY(hipMemsetAsync(S.st, 0, sizeof(FuseStats), ctx->stream));
    I64View V; V.A = dA; V.B = dA; V.c = c; V.ch = ch;
    hipLaunchKernelGGL(k_i64_stats_rows, dim3(r), dim3(256), 0, ctx->stream, V, r, c, S.st, S.rowFirst, S.rowLast);
    hipLaunchKernelGGL(k_i64_stats_cols, dim3((c + 255) / 256), dim3(256), 0, ctx->stream, V, r, c, S.colFirst, S.colLast);
    TRY(launch_weights(ctx, S, r, c, ch, dx, dy, force_corner));
    return finish_weights(ctx, S, r, c, info, h_ramps);
}

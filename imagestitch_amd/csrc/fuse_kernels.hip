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
#include <vector>
#include <string.h>

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

// one workgroup per ROI row: occupancy + quadrant counts + first/last valid column of the row
__global__ __launch_bounds__(256) void k_fuse_stats_rows(CanvasView V, int r, int c, FuseStats *st, int *rowFirst, int *rowLast)
{
    const int i = blockIdx.x;
    int first = 0x7fffffff, last = -1;
    unsigned valid = 0, qlo = 0, qhi = 0;          // > 0 counts left / right half
    const int c2 = c / 2;
    for (int j = threadIdx.x; j < c; j += 256) {
        if (V.a_valid(i, j)) {
            first = min(first, j); last = max(last, j);
            valid += V.ch;
            int pos = 0;
            for (int k = 0; k < V.ch; k++) pos += V.a_val(i, j, k) > 0;
            if (j < c2) qlo += pos; else qhi += pos;
        }
    }
    __shared__ int sf[256], sl[256];
    __shared__ unsigned sv[256], s0[256], s1[256];
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
        atomicAdd(&st->quad[top ? 0 : 1], (unsigned long long)s0[0]);   // TL / BL
        atomicAdd(&st->quad[top ? 3 : 2], (unsigned long long)s1[0]);   // TR / BR
    }
}

// one lane per ROI column: first/last valid row
__global__ __launch_bounds__(256) void k_fuse_stats_cols(CanvasView V, int r, int c, int *colFirst, int *colLast)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= c) return;
    int first = -1, last = -1;
    for (int i = 0; i < r; i++)
        if (V.a_valid(i, j)) { if (first < 0) first = i; last = i; }
    colFirst[j] = first; colLast[j] = last;
}

// the blend: writes the whole tile rectangle (outside the ROI: plain paste) and marks it valid
__global__ __launch_bounds__(256) void k_fuse_apply(uint8_t *pix, uint8_t *mask, int ccols, int ch,
                                                    const uint8_t *tile, int th, int tw, int y0, int x0,
                                                    int ry0, int rx0, int r, int c, int corner,
                                                    const float *wAr, const float *wAc, const float *wBr, const float *wBc)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= tw) return;
    const int cy = y0 + y, cx = x0 + x;
    const size_t co = (size_t)cy * ccols + cx;
    const int i = cy - ry0, j = cx - rx0;
    const bool in_roi = (i >= 0 && i < r && j >= 0 && j < c);
    if (in_roi) {
        float wA, wB;
        if (corner) { wB = wBr[i] * wBc[j]; wA = 1 - wB; }
        else { wA = wAr[i] * wAc[j]; wB = wBr[i] * wBc[j]; }
        const bool av = mask[co] != 0;
        for (int k = 0; k < ch; k++) {
            const int b = tile[((size_t)y * tw + x) * ch + k];
            const int a = av ? (int)pix[co * ch + k] : b;          // imageA[imageA < 0] = imageB[imageA < 0]
            double res = (double)wA * (double)a + (double)wB * (double)b;
            res = res < 0 ? 0 : res;
            res = res > 255 ? 255 : res;
            pix[co * ch + k] = (uint8_t)res;                       // np.uint8(): truncation
        }
    } else {
        for (int k = 0; k < ch; k++) pix[co * ch + k] = tile[((size_t)y * tw + x) * ch + k];
    }
    mask[co] = 1;
}

__global__ __launch_bounds__(256) void k_paste(uint8_t *pix, uint8_t *mask, int ccols, int ch,
                                               const uint8_t *tile, int th, int tw, int y0, int x0)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= tw) return;
    const size_t co = (size_t)(y0 + y) * ccols + (x0 + x);
    for (int k = 0; k < ch; k++) pix[co * ch + k] = tile[((size_t)y * tw + x) * ch + k];
    mask[co] = 1;
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
__global__ __launch_bounds__(256) void k_i64_apply(I64View V, int r, int c, int corner, const float *wAr, const float *wAc,
                                                   const float *wBr, const float *wBc, uint8_t *out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= c) return;
    float wA, wB;
    if (corner) { wB = wBr[i] * wBc[j]; wA = 1 - wB; }
    else { wA = wAr[i] * wAc[j]; wB = wBr[i] * wBc[j]; }
    for (int k = 0; k < V.ch; k++) {
        long long a = V.a_raw(i, j, k), b = V.b_val(i, j, k);
        if (a < 0) a = b;
        double res = (double)wA * (double)a + (double)wB * (double)b;
        res = res < 0 ? 0 : res;
        res = res > 255 ? 255 : res;
        out[((size_t)i * c + j) * V.ch + k] = (uint8_t)res;
    }
}

// ---- host: mode decision + separable ramps (ImageFusion.py:204-239 and :43-190) -----------------------
static inline int pywrap(int i, int n) { return i < 0 ? i + n : i; }

// returns 0 ok, -1 where the reference itself would raise (IndexError / ZeroDivisionError)
static int build_weights(int r, int c, int ch, int dx, int dy, const FuseStats &st,
                         const std::vector<int> &rowFirst, const std::vector<int> &rowLast,
                         const std::vector<int> &colFirst, const std::vector<int> &colLast,
                         std::vector<float> &wAr, std::vector<float> &wAc, std::vector<float> &wBr, std::vector<float> &wBc,
                         int *corner_out, int32_t *info)
{
    wAr.assign(r, 1.f); wBr.assign(r, 1.f); wAc.assign(c, 1.f); wBc.assign(c, 1.f);
    const double nel = (double)r * c * ch;
    int32_t inf[4] = {0, -1, 0, 0};
    *corner_out = 0;
    if ((double)st.valid / nel > 0.65) {
        if (c <= r) {                       // side-by-side strip: ramps along columns (float32 arithmetic)
            for (int i = 0; i < c; i++) {
                const float f = (dy >= 0) ? (float)i : (float)(c - i);
                wAc[c - i - 1] = ((wAc[c - i - 1] * f) * 1.0f) / (float)c;
                wBc[i] = ((wBc[i] * f) * 1.0f) / (float)c;
            }
        } else {                            // stacked strip: ramps along rows
            for (int i = 0; i < r; i++) {
                const float f = (dx <= 0) ? (float)i : (float)(r - i);
                wAr[i] = ((wAr[i] * f) * 1.0f) / (float)r;
                wBr[r - i - 1] = ((wBr[r - i - 1] * f) * 1.0f) / (float)r;
            }
        }
    } else {
        *corner_out = 1; inf[0] = 1;
        int index = 0;
        for (int q = 1; q < 4; q++) if (st.quad[q] < st.quad[index]) index = q;
        int rowIndex = 0, colIndex = 0;
        if (index == 2 || index == 3) {
            for (int j = 1; j < c; j++) {
                const int cj = c - j;
                if (index == 2) { if (colLast[cj] >= 0) rowIndex = colLast[cj] + 1; }
                else            { if (colFirst[cj] >= 0) rowIndex = colFirst[cj] - 1; }
                if (rowIndex != 0) break;
            }
            if (rowIndex >= r) return -1;
            const int rr = pywrap(rowIndex, r);
            if (rowLast[rr] >= 0) colIndex = rowLast[rr] + 1;
        } else {
            for (int j = 0; j < c; j++) {
                if (index == 0) { if (colFirst[j] >= 0) rowIndex = colFirst[j] - 1; }
                else            { if (colLast[j] >= 0) rowIndex = colLast[j] + 1; }
                if (rowIndex != 0) break;
            }
            if (rowIndex >= r) return -1;
            const int rr = pywrap(rowIndex, r);
            if (rowFirst[rr] >= 0) colIndex = rowFirst[rr] - 1;
        }
        inf[1] = index; inf[2] = rowIndex; inf[3] = colIndex;
        if (index == 2 || index == 1) {
            const int n = rowIndex + 1; int ri = rowIndex;
            for (int i = 0; i < n; i++) {
                if (ri == 0) ri = 1;
                const int idx = ri - i;
                if (idx >= r) return -1;
                wBr[pywrap(idx, r)] = (float)((double)(ri - i) * 1 / ri);
            }
        } else {
            int ri = rowIndex;
            for (int i = rowIndex; i < r; i++) {
                if (ri == 0) ri = 1;
                if (r - ri - 1 == 0) return -1;
                wBr[pywrap(i, r)] = (float)((double)(r - i - 1) * 1 / (r - ri - 1));
            }
        }
        if (index == 2 || index == 3) {
            const int n = colIndex + 1; int ci = colIndex;
            for (int i = 0; i < n; i++) {
                if (ci == 0) ci = 1;
                const int idx = ci - i;
                if (idx >= c) return -1;
                wBc[pywrap(idx, c)] = (float)((double)(ci - i) * 1 / ci);
            }
        } else {
            int ci = colIndex;
            for (int i = colIndex; i < c; i++) {
                if (ci == 0) ci = 1;
                if (c - ci - 1 == 0) return -1;
                wBc[pywrap(i, c)] = (float)((double)(c - i - 1) * 1 / (c - ci - 1));
            }
        }
    }
    if (info) for (int k = 0; k < 4; k++) info[k] = inf[k];
    return 0;
}

struct FuseScratch { FuseStats *st; int *rowFirst, *rowLast, *colFirst, *colLast; float *wAr, *wAc, *wBr, *wBc; };

static int fuse_scratch(vfsms_ctx *ctx, int r, int c, FuseScratch *S)
{
    S->st = (FuseStats *)ctx_arena_alloc(ctx, sizeof(FuseStats));
    S->rowFirst = (int *)ctx_arena_alloc(ctx, sizeof(int) * r); S->rowLast = (int *)ctx_arena_alloc(ctx, sizeof(int) * r);
    S->colFirst = (int *)ctx_arena_alloc(ctx, sizeof(int) * c); S->colLast = (int *)ctx_arena_alloc(ctx, sizeof(int) * c);
    S->wAr = (float *)ctx_arena_alloc(ctx, sizeof(float) * r); S->wBr = (float *)ctx_arena_alloc(ctx, sizeof(float) * r);
    S->wAc = (float *)ctx_arena_alloc(ctx, sizeof(float) * c); S->wBc = (float *)ctx_arena_alloc(ctx, sizeof(float) * c);
    if (!S->wBc) { vfsms_set_error("arena exhausted in fuse"); return VFSMS_ERR_CAPACITY; }
    return VFSMS_OK;
}

static int fetch_stats_and_weights(vfsms_ctx *ctx, const FuseScratch &S, int r, int c, int ch, int dx, int dy,
                                   int *corner, int32_t *info, float *h_ramps = nullptr, int force_corner = 0)
{
    FuseStats st;
    std::vector<int> rf(r), rl(r), cf(c), cl(c);
    HIP_TRY(hipMemcpyAsync(&st, S.st, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(rf.data(), S.rowFirst, sizeof(int) * r, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(rl.data(), S.rowLast, sizeof(int) * r, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(cf.data(), S.colFirst, sizeof(int) * c, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(cl.data(), S.colLast, sizeof(int) * c, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<float> wAr, wAc, wBr, wBc;
    if (force_corner) st.valid = 0;                      // getWeightsMatrix called directly: skip the occupancy test
    if (build_weights(r, c, ch, dx, dy, st, rf, rl, cf, cl, wAr, wAc, wBr, wBc, corner, info) != 0) {
        vfsms_set_error("fuse: degenerate corner geometry (the reference's getWeightsMatrix raises here)");
        return VFSMS_ERR_BAD_ARG;
    }
    HIP_TRY(hipMemcpyAsync(S.wAr, wAr.data(), sizeof(float) * r, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(S.wBr, wBr.data(), sizeof(float) * r, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(S.wAc, wAc.data(), sizeof(float) * c, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(S.wBc, wBc.data(), sizeof(float) * c, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));    // host vectors go out of scope
    if (h_ramps) {                                 // [wA_r | wB_r | wA_c | wB_c]
        memcpy(h_ramps, wAr.data(), sizeof(float) * r); memcpy(h_ramps + r, wBr.data(), sizeof(float) * r);
        memcpy(h_ramps + 2 * r, wAc.data(), sizeof(float) * c); memcpy(h_ramps + 2 * r + c, wBc.data(), sizeof(float) * c);
    }
    return VFSMS_OK;
}

int canvas_paste_device(vfsms_ctx *ctx, CanvasRec *cv, const uint8_t *d_tile, int h, int w, int y0, int x0)
{
    hipLaunchKernelGGL(k_paste, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, cv->pix, cv->mask, cv->cols, cv->ch,
                       d_tile, h, w, y0, x0);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

int canvas_fuse_device(vfsms_ctx *ctx, CanvasRec *cv, const uint8_t *d_tile, int h, int w, int y0, int x0,
                       int ry0, int rx0, int ry1, int rx1, int dx, int dy, int32_t *info)
{
    const int r = ry1 - ry0, c = rx1 - rx0;
    if (r <= 0 || c <= 0) return canvas_paste_device(ctx, cv, d_tile, h, w, y0, x0);
    ProfScope ps(ctx, "fuse");
    FuseScratch S;
    TRY(fuse_scratch(ctx, r, c, &S));
    HIP_TRY(hipMemsetAsync(S.st, 0, sizeof(FuseStats), ctx->stream));
    CanvasView V;
    V.pix = cv->pix; V.mask = cv->mask; V.ccols = cv->cols; V.ch = cv->ch; V.ry0 = ry0; V.rx0 = rx0;
    V.tile = d_tile; V.tw = w; V.ty0 = ry0 - y0; V.tx0 = rx0 - x0;
    hipLaunchKernelGGL(k_fuse_stats_rows, dim3(r), dim3(256), 0, ctx->stream, V, r, c, S.st, S.rowFirst, S.rowLast);
    hipLaunchKernelGGL(k_fuse_stats_cols, dim3((c + 255) / 256), dim3(256), 0, ctx->stream, V, r, c, S.colFirst, S.colLast);
    int corner = 0;
    TRY(fetch_stats_and_weights(ctx, S, r, c, cv->ch, dx, dy, &corner, info));
    hipLaunchKernelGGL(k_fuse_apply, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, cv->pix, cv->mask, cv->cols, cv->ch,
                       d_tile, h, w, y0, x0, ry0, rx0, r, c, corner, S.wAr, S.wAc, S.wBr, S.wBc);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

// A, B: device int64 [r][c][ch]; out: device u8
int fuse_i64_device(vfsms_ctx *ctx, const long long *dA, const long long *dB, int r, int c, int ch, int dx, int dy,
                    uint8_t *d_out, int32_t *info)
{
    FuseScratch S;
    TRY(fuse_scratch(ctx, r, c, &S));
    HIP_TRY(hipMemsetAsync(S.st, 0, sizeof(FuseStats), ctx->stream));
    I64View V; V.A = dA; V.B = dB; V.c = c; V.ch = ch;
    hipLaunchKernelGGL(k_i64_stats_rows, dim3(r), dim3(256), 0, ctx->stream, V, r, c, S.st, S.rowFirst, S.rowLast);
    hipLaunchKernelGGL(k_i64_stats_cols, dim3((c + 255) / 256), dim3(256), 0, ctx->stream, V, r, c, S.colFirst, S.colLast);
    int corner = 0;
    TRY(fetch_stats_and_weights(ctx, S, r, c, ch, dx, dy, &corner, info));
    hipLaunchKernelGGL(k_i64_apply, dim3((c + 255) / 256, r), dim3(256), 0, ctx->stream, V, r, c, corner,
                       S.wAr, S.wAc, S.wBr, S.wBc, d_out);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
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
    int corner = 0;
    TRY(fetch_stats_and_weights(ctx, S, r, c, ch, dx, dy, &corner, info, h_ramps, force_corner));
    return VFSMS_OK;
}

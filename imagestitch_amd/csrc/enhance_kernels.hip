// enhance_kernels.hip -- histogram equalisation and CLAHE of 8-bit ROIs on gfx950, batched over jobs.
//
// The reference's optional pre-enhancement (Method.isEnhance / isClahe / clipLimit / tileSize, ImageUtility.py:46-50):
//   Stitcher.py:269-276 (whole tiles, calculateOffsetForFeatureSearch) and Stitcher.py:327-334 (ROI strips of the incremental
//   search) call cv2.createCLAHE(clipLimit, (tileSize, tileSize)).apply(img) or cv2.equalizeHist(img) before detectAndDescribe.
// Semantics are OpenCV 3.3.1's (imgproc/src/histogram.cpp equalizeHist; imgproc/src/clahe.cpp CLAHE_Impl::apply, CV_8UC1):
//   equalizeHist: hist -> first non-empty bin i0 -> lut[i] = saturate(cvRound(sum_{i0<j<=i} hist[j] * (255.f / (total - hist[i0]))))
//   CLAHE: tilesX x tilesY grid over the image extended bottom/right by BORDER_REFLECT_101 to a multiple of the grid, per tile
//          histogram clipped at int(clipLimit * tileArea / 256) (>= 1), excess spread evenly + the first `residual` bins, LUT =
//          saturate(cvRound(cumsum * (255.f / tileArea))), then per pixel the bilinear blend of the four neighbouring tile LUTs
//          in float, in upstream's operation order (-ffp-contract=off).
// Three launches per batch: tile histograms (LDS-private, one global atomic per non-empty bin), LUTs (one workgroup per tile:
// block reduction + scan), apply (HBM-bound: 1 B/px in, 1 B/px out, LUTs from L2).
#include "common.h"
#include <math.h>
#include <algorithm>

#define GAS __attribute__((address_space(1)))

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}
__device__ __forceinline__ uint8_t sat_u8f(float v)
{
    const int iv = (int)rintf(v);
    return (uint8_t)(iv < 0 ? 0 : iv > 255 ? 255 : iv);
}

// grid: (blocks per tile, tiles, jobs).  mode 1: equalizeHist (one "tile" = the image, no extension); mode 2: CLAHE
__global__ __launch_bounds__(256) void k_enh_hist(const EnhJob *jobs, int mode, int tilesX, int tilesY)
{
    const EnhJob J = jobs[blockIdx.z];
    __shared__ int hs[256];
    hs[threadIdx.x] = 0;
    __syncthreads();
    int x0 = 0, y0 = 0, tw = J.w, th = J.h;
    if (mode == 2) {
        tw = J.ew / tilesX; th = J.eh / tilesY;
        x0 = (blockIdx.y % tilesX) * tw; y0 = (blockIdx.y / tilesX) * th;
    }
    const uint8_t GAS *src = (const uint8_t GAS *)J.src;
    for (int y = blockIdx.x; y < th; y += gridDim.x) {
        const int sy = reflect101(y0 + y, J.h);
        const uint8_t GAS *row = src + (size_t)sy * J.stride;
        for (int x = threadIdx.x; x < tw; x += 256) atomicAdd(&hs[row[reflect101(x0 + x, J.w)]], 1);
    }
    __syncthreads();
    const int v = hs[threadIdx.x];
    if (v) atomicAdd(&J.hist[blockIdx.y * 256 + threadIdx.x], v);
}

__device__ __forceinline__ int block_incl_scan_256(int v, int *wsum)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int n = __shfl_up(incl, d, 64); if (lane >= d) incl += n; }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int off = 0;
    for (int k = 0; k < wid; k++) off += wsum[k];
    __syncthreads();
    return incl + off;
}

// grid: (tiles, jobs), 256 threads = 256 bins
__global__ __launch_bounds__(256) void k_enh_lut(const EnhJob *jobs, int mode, int tilesX, int tilesY, double clipLimitD)
{
    const EnhJob J = jobs[blockIdx.y];
    __shared__ int wsum[4];
    __shared__ int red[2];
    const int i = threadIdx.x;
    int h = J.hist[blockIdx.x * 256 + i];
    uint8_t out;
    if (mode == 1) {
        // first non-empty bin
        if (i == 0) red[0] = 256;
        __syncthreads();
        if (h) atomicMin(&red[0], i);
        __syncthreads();
        const int i0 = red[0];
        const int total = J.h * J.w;
        const int incl = block_incl_scan_256(h, wsum);
        if (i == i0) red[1] = incl;                       // == hist[i0] (nothing below it)
        __syncthreads();
        const int h0 = red[1];
        if (h0 == total) out = (uint8_t)i0;               // one grey level only: dst.setTo(i0)
        else {
            const float scale = (256 - 1.f) / (float)(total - h0);
            out = i <= i0 ? (uint8_t)0 : sat_u8f((float)(incl - h0) * scale);
        }
    } else {
        const int tw = J.ew / tilesX, th = J.eh / tilesY;
        const int tileSizeTotal = tw * th;
        const float lutScale = (float)(256 - 1) / (float)tileSizeTotal;
        int clipLimit = 0;
        if (clipLimitD > 0.0) {
            clipLimit = (int)(clipLimitD * tileSizeTotal / 256);
            clipLimit = max(clipLimit, 1);
        }
        if (clipLimit > 0) {
            const int excess = max(h - clipLimit, 0);
            const int tot_excess = block_incl_scan_256(excess, wsum);
            if (i == 255) red[0] = tot_excess;
            __syncthreads();
            const int clipped = red[0];
            const int redistBatch = clipped / 256, residual = clipped - redistBatch * 256;
            h = min(h, clipLimit) + redistBatch + (i < residual ? 1 : 0);
        }
        const int sum = block_incl_scan_256(h, wsum);
        out = sat_u8f((float)sum * lutScale);
    }
    J.lut[blockIdx.x * 256 + i] = out;
}

// grid: (ceil(w / 256), h, jobs)
__global__ __launch_bounds__(256) void k_enh_apply(const EnhJob *jobs, int mode, int tilesX, int tilesY)
{
    const EnhJob J = jobs[blockIdx.z];
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= J.w || y >= J.h) return;
    const int v = ((const uint8_t GAS *)J.src)[(size_t)y * J.stride + x];
    const uint8_t GAS *lut = (const uint8_t GAS *)J.lut;
    uint8_t o;
    if (mode == 1) o = lut[v];
    else {
        const int tw = J.ew / tilesX, th = J.eh / tilesY;
        const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        ty1 = max(ty1, 0); ty2 = min(ty2, tilesY - 1);
        const float txf = (float)x * inv_tw - 0.5f;
        int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
        const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
        tx1 = max(tx1, 0); tx2 = min(tx2, tilesX - 1);
        const uint8_t GAS *p1 = lut + (size_t)ty1 * tilesX * 256, *p2 = lut + (size_t)ty2 * tilesX * 256;
        const int ind1 = tx1 * 256 + v, ind2 = tx2 * 256 + v;
        const float res = ((float)p1[ind1] * xa1 + (float)p1[ind2] * xa) * ya1 + ((float)p2[ind1] * xa1 + (float)p2[ind2] * xa) * ya;
        o = sat_u8f(res);
    }
    ((uint8_t GAS *)J.dst)[(size_t)y * J.w + x] = o;
}

size_t enhance_scratch_bytes(int h, int w, int mode, int tiles)
{
    const int nt = mode == 2 ? tiles * tiles : 1;
    return (((size_t)h * w + 255) & ~(size_t)255) + (((size_t)nt * 256 * 4 + 255) & ~(size_t)255) + (((size_t)nt * 256 + 255) & ~(size_t)255);
}

// Fill the device-side fields of one job: the enhanced image (contiguous h x w), its histograms and LUTs come out of the arena.
int enhance_carve(vfsms_ctx *ctx, EnhJob *J, const uint8_t *src, int stride, int h, int w, int mode, int tiles)
{
    const int nt = mode == 2 ? tiles * tiles : 1;
    J->src = src; J->stride = stride; J->h = h; J->w = w; J->eh = h; J->ew = w;
    if (mode == 2 && !(w % tiles == 0 && h % tiles == 0)) { J->eh = h + (tiles - h % tiles); J->ew = w + (tiles - w % tiles); }
    J->dst = (uint8_t *)ctx_arena_alloc(ctx, (size_t)h * w);
    J->hist = (int *)ctx_arena_alloc(ctx, (size_t)nt * 256 * sizeof(int));
    J->lut = (uint8_t *)ctx_arena_alloc(ctx, (size_t)nt * 256);
    if (!J->dst || !J->hist || !J->lut) { vfsms_set_error("arena exhausted (enhancement)"); return VFSMS_ERR_CAPACITY; }
    return VFSMS_OK;
}

// h_jobs: carved jobs (host copy); d_jobs: the same array on the device.  Stream-ordered, no host sync.
int launch_enhance(vfsms_ctx *ctx, const EnhJob *d_jobs, const EnhJob *h_jobs, int n, int mode, double clip_limit, int tiles)
{
    if (n <= 0 || mode == 0) return VFSMS_OK;
    if (mode != 1 && mode != 2) { vfsms_set_error("enhance: mode must be 0 (none), 1 (equalizeHist) or 2 (CLAHE)"); return VFSMS_ERR_BAD_ARG; }
    if (mode == 2 && (tiles < 1 || tiles > 64)) { vfsms_set_error("enhance: CLAHE tile grid must be 1..64"); return VFSMS_ERR_BAD_ARG; }
    ProfScope ps(ctx, "enhance");
    const int nt = mode == 2 ? tiles * tiles : 1;
    int maxh = 0, maxw = 0;
    for (int k = 0; k < n; k++) {
        HIP_TRY(hipMemsetAsync(h_jobs[k].hist, 0, (size_t)nt * 256 * sizeof(int), ctx->stream));
        maxh = std::max(maxh, h_jobs[k].h); maxw = std::max(maxw, h_jobs[k].w);
    }
    const int th = mode == 2 ? (maxh + tiles) / tiles : maxh;
    const int bpt = std::max(1, std::min(th, mode == 2 ? 16 : 256));
    hipLaunchKernelGGL(k_enh_hist, dim3(bpt, nt, n), dim3(256), 0, ctx->stream, d_jobs, mode, tiles, tiles);
    hipLaunchKernelGGL(k_enh_lut, dim3(nt, n), dim3(256), 0, ctx->stream, d_jobs, mode, tiles, tiles, clip_limit);
    hipLaunchKernelGGL(k_enh_apply, dim3((maxw + 255) / 256, maxh, n), dim3(256), 0, ctx->stream, d_jobs, mode, tiles, tiles);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

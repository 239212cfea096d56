// orb_kernels.hip -- ORB detect + describe for gfx950, batched over ROI records.
//
// Replaces cv2.ORB_create(5000, 1.2, 8, 31, 0, 2, HARRIS_SCORE, 31, 20).detectAndCompute (ImageUtility.py:260,262) and the
// DLL twin detectAndDescribeByOrb (appendix/myGpuFeatures.cpp:106-146): OpenCV 3.3.1 features2d/src/{orb,fast,fast_score}.cpp
// semantics (DESIGN.md section 3 lists the two stated deviations from upstream: sampling pattern from upstream's
// makeRandomPattern, retainBest in detection order).  Everything here is integer or fixed-point except the Harris score and the
// pattern rotation, whose float operation order is explicit (-ffp-contract=off).
//
// Pipeline per batch (no host sync): fixed-point bilinear pyramid (level by level) -> FAST-9/16 score map (all levels, one
// launch) -> 3x3 NMS + border filter + per-level score histogram -> histogram threshold (= "keep everything >= the n-th best",
// exact for integer scores) -> ordered compaction + Harris response -> per-level rank-by-counting selection + intensity
// centroid angle -> 7x7 fixed-point Gaussian blur (LDS tile) -> rotated-BRIEF bytes.  All of it is byte/int streaming work,
// HBM/L2-bound; nothing is shaped into a GEMM.
#include "common.h"
#include "detmath.h"
#include "orb_pattern31.h"
#include <math.h>
#include <float.h>
#include <string.h>

#define GAS __attribute__((address_space(1)))
typedef GAS const uint8_t *g_cu8;
typedef GAS uint8_t *g_u8;

__device__ __forceinline__ int cv_round_f(float v) { return (int)rintf(v); }
__device__ __forceinline__ int cv_floor_d(double v) { return (int)floor(v); }
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}
__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// ---- pyramid: resize(8U, INTER_LINEAR) in 11-bit fixed point, one lane per destination pixel ---------------------------------
__global__ __launch_bounds__(256) void k_orb_resize(const OrbDev *rois, int level)
{
    const OrbDev &R = rois[blockIdx.z];
    const int dw = R.lw[level], dh = R.lh[level], sw = R.lw[level - 1], sh = R.lh[level - 1];
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw || dy >= dh) return;
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_d(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    bool single = false;
    if (sx + 1 >= sw) { single = true; if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
    const int a0 = (short)cv_round_f((1.f - fx) * 2048), a1 = (short)cv_round_f(fx * 2048);
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_d(fy);
    fy -= sy;
    const int b0 = (short)cv_round_f((1.f - fy) * 2048), b1 = (short)cv_round_f(fy * 2048);
    const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
    g_cu8 S0 = (g_cu8)R.lv[level - 1] + (size_t)sy0 * R.ls[level - 1];
    g_cu8 S1 = (g_cu8)R.lv[level - 1] + (size_t)sy1 * R.ls[level - 1];
    int r0, r1;
    if (!single) { r0 = S0[sx] * a0 + S0[sx + 1] * a1; r1 = S1[sx] * a0 + S1[sx + 1] * a1; }
    else { r0 = S0[sx] * 2048; r1 = S1[sx] * 2048; }
    ((g_u8)R.lv[level])[(size_t)dy * R.ls[level] + dx] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
}

// ---- FAST-9/16: corner test + cornerScore<16>, all levels in one launch (blockIdx.z = roi * nlevels + level) --------------------
__device__ __forceinline__ int corner_score16(const int *d /* 25 differences v - ring[k] */, int threshold)
{
    int a0 = threshold;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        int a = min(d[k + 1], d[k + 2]);
        a = min(a, d[k + 3]);
        if (a <= a0) continue;
        a = min(a, d[k + 4]); a = min(a, d[k + 5]); a = min(a, d[k + 6]); a = min(a, d[k + 7]); a = min(a, d[k + 8]);
        a0 = max(a0, min(a, d[k]));
        a0 = max(a0, min(a, d[k + 9]));
    }
    int b0 = -a0;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        int b = max(d[k + 1], d[k + 2]);
        b = max(b, d[k + 3]); b = max(b, d[k + 4]); b = max(b, d[k + 5]);
        if (b >= b0) continue;
        b = max(b, d[k + 6]); b = max(b, d[k + 7]); b = max(b, d[k + 8]);
        b0 = min(b0, max(b, d[k]));
        b0 = min(b0, max(b, d[k + 9]));
    }
    return -b0 - 1;
}

// ---- fused FAST score + 3x3 NMS + border rule + score histogram: one 64 x 16 pixel tile per workgroup ---------------------------------
// The tile plus a 4 px halo (1 for the NMS neighbourhood, 3 for the Bresenham ring) is staged in LDS once; scores of the tile and its
// 1 px halo are computed from LDS and never leave it; only the NMS survivors' byte map goes to HBM.  The histogram of surviving scores
// is accumulated in LDS and flushed with at most 256 global atomics per workgroup (scores cluster on few values: per-pixel global
// atomics on the same few addresses were the bulk of this stage).
#define FT_W 64
#define FT_H 16
// blockIdx.x walks the tiles of level 0, then level 1, ... of ONE ROI shape (first[l] = tiles before level l): a grid cut from the level-0
// size for every level dispatched 2.4 times the workgroups the pyramid has tiles
struct OrbPlan { int first[VFSMS_ORB_MAX_LEVELS + 1]; int tiles_x[VFSMS_ORB_MAX_LEVELS]; };
__device__ __forceinline__ void orb_plan_tile(const OrbPlan &plan, int nlevels, int &level, int &tx, int &ty)
{
    int t = blockIdx.x;
    level = 0;
    while (level + 1 < nlevels && t >= plan.first[level + 1]) level++;
    t -= plan.first[level];
    ty = t / plan.tiles_x[level]; tx = t - ty * plan.tiles_x[level];
}
__global__ __launch_bounds__(256) void k_orb_fast_nms(const OrbDev *rois, int nlevels, int threshold, int edge, OrbPlan plan)
{
    const OrbDev &R = rois[blockIdx.y];
    int level, tile_x, tile_y;
    orb_plan_tile(plan, nlevels, level, tile_x, tile_y);
    const int w = R.lw[level], h = R.lh[level], st = R.ls[level];
    const int x0 = tile_x * FT_W, y0 = tile_y * FT_H;
    if (x0 >= w || y0 >= h) return;
    __shared__ uint8_t img[FT_H + 8][FT_W + 8];
    __shared__ uint8_t sc[FT_H + 2][FT_W + 2 + 2];
    __shared__ int hist[256];
    const int tid = threadIdx.x;
    hist[tid] = 0;
    g_cu8 src = (g_cu8)R.lv[level];
    for (int idx = tid; idx < (FT_H + 8) * (FT_W + 8); idx += 256) {
        const int r = idx / (FT_W + 8), c = idx - r * (FT_W + 8);
        const int gy = y0 - 4 + r, gx = x0 - 4 + c;
        img[r][c] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? src[(size_t)gy * st + gx] : (uint8_t)0;
    }
    __syncthreads();
    // Pass A: the 9-of-16 segment test for every pixel of the tile + its 1 px halo; pixels that pass are queued.  Pass B: cornerScore
    // for the queue only, all lanes busy.  (Evaluated inline, a wave ran the ~150-instruction score whenever ONE of its 64 pixels
    // was a corner candidate -- nearly always -- although only about one pixel in ten is.)
    __shared__ unsigned short cq[(FT_H + 2) * (FT_W + 2)];
    __shared__ int cqn;
    if (tid == 0) cqn = 0;
    const int ox[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    const int oy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    __syncthreads();
    for (int idx = tid; idx < (FT_H + 2) * (FT_W + 2); idx += 256) {
        const int r = idx / (FT_W + 2), c = idx - r * (FT_W + 2);
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        bool cand = false;
        if (gy >= 3 && gy < h - 3 && gx >= 3 && gx < w - 3) {
            const uint8_t *p = &img[r + 3][c + 3];
            const int v = p[0];
            unsigned dark = 0, bright = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int x = p[oy[k] * (FT_W + 8) + ox[k]];
                dark |= (unsigned)(x < v - threshold) << k;
                bright |= (unsigned)(x > v + threshold) << k;
            }
            // a circular run of >= 9: AND of the mask with its 8 rotations
            unsigned md = dark | (dark << 16), mb = bright | (bright << 16);
            unsigned rd = md, rb = mb;
#pragma unroll
            for (int q = 1; q < 9; q++) { rd &= md >> q; rb &= mb >> q; }
            cand = ((rd & 0xffffu) | (rb & 0xffffu)) != 0;
        }
        sc[r][c] = 0;
        if (cand) cq[atomicAdd(&cqn, 1)] = (unsigned short)idx;
    }
    __syncthreads();
    const int ncand = cqn;
    for (int e = tid; e < ncand; e += 256) {
        const int idx = cq[e];
        const int r = idx / (FT_W + 2), c = idx - r * (FT_W + 2);
        const uint8_t *p = &img[r + 3][c + 3];
        const int v = p[0];
        int d[25];
#pragma unroll
        for (int k = 0; k < 16; k++) d[k] = v - (int)p[oy[k] * (FT_W + 8) + ox[k]];
#pragma unroll
        for (int k = 16; k < 25; k++) d[k] = d[k - 16];
        sc[r][c] = (uint8_t)(corner_score16(d, threshold) & 0xff);
    }
    __syncthreads();
    g_u8 nm = (g_u8)R.nms[level];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = (tid >> 6) + 4 * k, c = tid & 63;
        const int i = y0 + r, j = x0 + c;
        if (i >= h || j >= w) continue;
        const int s = sc[r + 1][c + 1];
        int keep = 0;
        if (s && i >= 1 && i < h - 1 && j >= 1 && j < w - 1) {
            if (s > sc[r + 1][c + 2] && s > sc[r + 1][c] && s > sc[r][c] && s > sc[r][c + 1] && s > sc[r][c + 2] &&
                s > sc[r + 2][c] && s > sc[r + 2][c + 1] && s > sc[r + 2][c + 2])
                if (j >= edge && j < w - edge && i >= edge && i < h - edge) keep = s;
        }
        nm[(size_t)i * w + j] = (uint8_t)keep;
        if (keep) atomicAdd(&hist[keep], 1);
    }
    __syncthreads();
    if (hist[tid]) atomicAdd(&R.hist[level * 256 + tid], hist[tid]);
}

// zero the per-ROI histograms and counters of a batch in one launch
__global__ __launch_bounds__(256) void k_orb_clear(const OrbDev *rois, int nlevels)
{
    const OrbDev &R = rois[blockIdx.x];
    for (int i = threadIdx.x; i < 256 * nlevels; i += 256) R.hist[i] = 0;
    if (threadIdx.x < 64) R.counters[threadIdx.x] = 0;
}

// keep every keypoint whose FAST score >= the n-th best (KeyPointsFilter::retainBest, HARRIS_SCORE keeps 2 * quota here)
__global__ void k_orb_threshold(const OrbDev *rois, int nrois, int nlevels, const OrbTables *T)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nrois * nlevels) return;
    const OrbDev &R = rois[t / nlevels];
    const int level = t % nlevels;
    const int n = 2 * T->nfeat[level];
    int acc = 0, thr = 1;
    for (int s = 255; s >= 1; s--) {
        acc += R.hist[level * 256 + s];
        if (acc >= n && n > 0) { thr = s; break; }
    }
    if (n == 0) thr = 256;                              // retainBest(0) keeps nothing
    R.thr1[level] = thr;
}

// ---- ordered compaction (row-major) of survivors + Harris response: one 1024-lane workgroup per (roi, level) ------------------------
__device__ __forceinline__ float harris_response(g_cu8 img, int st, int x0, int y0)
{
    const int r = 3, bs = 7;
    const float scale = 1.f / ((1 << 2) * bs * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    g_cu8 ptr0 = img + (size_t)(y0 - r) * st + x0 - r;
    int a = 0, b = 0, c = 0;
    for (int k = 0; k < bs * bs; k++) {
        g_cu8 p = ptr0 + (k / bs) * st + (k % bs);
        const int Ix = (p[1] - p[-1]) * 2 + (p[-st + 1] - p[-st - 1]) + (p[st + 1] - p[st - 1]);
        const int Iy = (p[st] - p[-st]) * 2 + (p[st - 1] - p[-st - 1]) + (p[st + 1] - p[-st + 1]);
        a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
    }
    return ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

// Ordered (row-major) compaction of the survivors in three fully parallel steps.  A level is cut into <= ORB_CHUNKS chunks of whole
// 4096-pixel blocks; the chunk counts and then the chunk offsets live in the level's score histogram, which k_orb_threshold has
// finished reading by then (first ORB_CHUNKS bins):   k_orb_count (count per chunk)  ->  k_orb_chunk_scan (exclusive scan, n1)  ->  k_orb_scatter.
#define ORB_CHUNKS 32
__device__ __forceinline__ long long orb_chunk_size(long long total)
{
    const long long blocks = (total + 4095) / 4096;
    return ((blocks + ORB_CHUNKS - 1) / ORB_CHUNKS) * 4096;
}

__global__ __launch_bounds__(1024) void k_orb_count(const OrbDev *rois, int nlevels)
{
    const OrbDev &R = rois[blockIdx.y / nlevels];
    const int level = blockIdx.y % nlevels;
    const long long total = (long long)R.lw[level] * R.lh[level];
    const long long cs = orb_chunk_size(total);
    const long long lo = (long long)blockIdx.x * cs, hi = min(total, lo + cs);
    const int thr = R.thr1[level];
    g_cu8 nm = (g_cu8)R.nms[level];
    int cnt = 0;
    for (long long p = lo + threadIdx.x; p < hi; p += 1024) { const int v = nm[p]; cnt += (v >= thr && v > 0) ? 1 : 0; }
    __shared__ int wsum[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int incl = wave_incl_scan(cnt);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < 16; k++) t += wsum[k];
        R.hist[level * 256 + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void k_orb_chunk_scan(const OrbDev *rois, int nlevels)
{
    const OrbDev &R = rois[blockIdx.x / nlevels];
    const int level = blockIdx.x % nlevels;
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = threadIdx.x < ORB_CHUNKS ? R.hist[level * 256 + threadIdx.x] : 0;
    const int incl = wave_incl_scan(c);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int off = incl - c;
    for (int k = 0; k < wid; k++) off += wsum[k];
    if (threadIdx.x < ORB_CHUNKS) R.hist[level * 256 + threadIdx.x] = off;
    if (threadIdx.x == 255) {
        const int total = off + c;
        if (total > R.cap1) R.counters[2] = 1;
        R.n1[level] = min(total, R.cap1);
    }
}

__global__ __launch_bounds__(1024) void k_orb_scatter(const OrbDev *rois, int nlevels)
{
    const OrbDev &R = rois[blockIdx.y / nlevels];
    const int level = blockIdx.y % nlevels;
    const int w = R.lw[level], h = R.lh[level];
    const long long total = (long long)w * h;
    const long long cs = orb_chunk_size(total);
    const long long lo = (long long)blockIdx.x * cs, hi = min(total, lo + cs);
    if (lo >= hi) return;
    const int thr = R.thr1[level];
    g_cu8 nm = (g_cu8)R.nms[level];
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = R.hist[level * 256 + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int *out_xy = R.k1_xy + (size_t)level * R.cap1 * 2;
    for (long long base = lo; base < hi; base += 4096) {        // 4 consecutive pixels per lane
        const long long p0 = base + (long long)threadIdx.x * 4;
        int f[4], cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { const long long p = p0 + q; f[q] = (p < hi && nm[p] >= thr && nm[p] > 0) ? 1 : 0; cnt += f[q]; }
        const int incl = wave_incl_scan(cnt);
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int off0 = carry + incl - cnt;
        for (int k = 0; k < wid; k++) off0 += wsum[k];
        int off = off0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (f[q]) {
                const long long p = p0 + q;
                const int y = (int)(p / w), x = (int)(p - (long long)y * w);
                if (off < R.cap1) { out_xy[2 * off] = x; out_xy[2 * off + 1] = y; }
                off++;
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = off0 + cnt;
        __syncthreads();
    }
}

// Harris response of the survivors, one thread each (it used to run inside the single-workgroup compaction loop, one lane at a time)
__global__ __launch_bounds__(256) void k_orb_harris(const OrbDev *rois, int nlevels)
{
    const OrbDev &R = rois[blockIdx.y / nlevels];
    const int level = blockIdx.y % nlevels;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= R.n1[level]) return;
    const int *xy = R.k1_xy + (size_t)level * R.cap1 * 2;
    R.k1_resp[(size_t)level * R.cap1 + idx] = harris_response((g_cu8)R.lv[level], R.ls[level], xy[2 * idx], xy[2 * idx + 1]);
}

// ---- second retainBest (quota) by Harris response, IC angle, ordered compaction: one workgroup per (roi, level) ----------------------
__global__ __launch_bounds__(1024) void k_orb_select2(const OrbDev *rois, int nlevels, const OrbTables *T)
{
    const OrbDev &R = rois[blockIdx.x / nlevels];
    const int level = blockIdx.x % nlevels;
    const int n1 = R.n1[level];
    const int quota = T->nfeat[level];
    const int *xy = R.k1_xy + (size_t)level * R.cap1 * 2;
    const float *resp = R.k1_resp + (size_t)level * R.cap1;
    __shared__ float tile[1024];
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int *o_xy = R.k2_xy + (size_t)level * R.cap2 * 2;
    float *o_r = R.k2_resp + (size_t)level * R.cap2;
    for (int base = 0; base < n1; base += 1024) {
        const int idx = base + threadIdx.x;
        const float me = idx < n1 ? resp[idx] : 0.f;
        int greater = 0;
        for (int tb = 0; tb < n1; tb += 1024) {
            __syncthreads();
            if (tb + (int)threadIdx.x < n1) tile[threadIdx.x] = resp[tb + threadIdx.x];
            __syncthreads();
            const int lim = min(1024, n1 - tb);
            if (idx < n1)
                for (int k = 0; k < lim; k++) greater += tile[k] > me ? 1 : 0;
        }
        // keep iff fewer than `quota` responses are strictly greater (== response >= the quota-th best); all kept if n1 <= quota
        const int keep = (idx < n1 && (n1 <= quota || greater < quota) && quota > 0) ? 1 : 0;
        const int incl = wave_incl_scan(keep);
        __syncthreads();
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int off = carry + incl - keep;
        for (int k = 0; k < wid; k++) off += wsum[k];
        if (keep) {
            const int x = xy[2 * idx], y = xy[2 * idx + 1];
            if (off < R.cap2) {
                o_xy[2 * off] = x; o_xy[2 * off + 1] = y; o_r[off] = me;         // the angle follows in k_orb_angle
            } else R.counters[2] = 1;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = off + keep;
        __syncthreads();
    }
    if (threadIdx.x == 0) R.n2[level] = min(carry, R.cap2);
}

// ---- ICAngles: intensity centroid over the circular patch (umax table), one WAVE per keypoint ---------------------------------------
// The moments are integer sums, so any summation order gives upstream's bits: lanes 0-31 / 32-63 take two rows of the disc per
// trip (rows are at most 31 pixels wide), one byte gather each, and a wave reduction finishes.  (Inside k_orb_select2 one lane
// walked the ~700 pixels of a disc with a dependent wait per load: 0.5 ms per launch.)
__global__ __launch_bounds__(256) void k_orb_angle(const OrbDev *rois, int nlevels, const OrbTables *T)
{
    const OrbDev &R = rois[blockIdx.y / nlevels];
    const int level = blockIdx.y % nlevels;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= R.n2[level]) return;
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int *xy = R.k2_xy + (size_t)level * R.cap2 * 2;
    const int x = xy[2 * idx], y = xy[2 * idx + 1];
    const int st = R.ls[level], hp = T->half_patch;
    g_cu8 center = (g_cu8)R.lv[level] + (size_t)y * st + x;
    int m_01 = 0, m_10 = 0;
    for (int v0 = -hp; v0 <= hp; v0 += 2) {
        const int v = v0 + half;
        if (v > hp) continue;
        const int d = T->umax[v < 0 ? -v : v];
        const int u = l32 - d;
        if (u > d) continue;
        const int val = center[u + v * st];
        m_10 += u * val; m_01 += v * val;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m_10 += __shfl_xor(m_10, o, 64); m_01 += __shfl_xor(m_01, o, 64); }
    if (lane != 0) return;
    // fastAtan2((float)m_01, (float)m_10)
    const float yy = (float)m_01, xx = (float)m_10;
    const float sc = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * sc, p3 = -0.3258083974640975f * sc, p5 = 0.1555786518463281f * sc, p7 = -0.04432655554792128f * sc;
    const float ax = fabsf(xx), ay = fabsf(yy);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (xx < 0) a = 180.f - a;
    if (yy < 0) a = 360.f - a;
    (R.k2_angle + (size_t)level * R.cap2)[idx] = a;
}


// ---- GaussianBlur(7x7, sigma 2, REFLECT_101) as the 8-bit fixed-point separable filter: 32x32 tile + 3 px halo in LDS ---------------
__global__ __launch_bounds__(256) void k_orb_blur(const OrbDev *rois, int nlevels, const OrbTables *T, OrbPlan plan)
{
    const OrbDev &R = rois[blockIdx.y];
    int level, tile_x, tile_y;
    orb_plan_tile(plan, nlevels, level, tile_x, tile_y);
    const int w = R.lw[level], h = R.lh[level], st = R.ls[level];
    const int x0 = tile_x * 32, y0 = tile_y * 32;
    if (x0 >= w || y0 >= h) return;
    __shared__ uint8_t src[38][40];
    __shared__ int rowp[38][32];
    g_cu8 img = (g_cu8)R.lv[level];
    for (int t = threadIdx.x; t < 38 * 38; t += 256) {
        const int ty = t / 38, tx = t % 38;
        src[ty][tx] = img[(size_t)reflect101(y0 + ty - 3, h) * st + reflect101(x0 + tx - 3, w)];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 38 * 32; t += 256) {
        const int ty = t / 32, tx = t % 32;
        int s = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) s += T->kf[k] * src[ty][tx + k];
        rowp[ty][tx] = s;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 32 * 32; t += 256) {
        const int ty = t / 32, tx = t % 32;
        if (y0 + ty < h && x0 + tx < w) {
            int s = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) s += T->kf[k] * rowp[ty + k][tx];
            s = (s + (1 << 15)) >> 16;
            ((g_u8)R.bl[level])[(size_t)(y0 + ty) * w + x0 + tx] = (uint8_t)(s < 0 ? 0 : s > 255 ? 255 : s);
        }
    }
}

// ---- rotated BRIEF: 32 lanes per keypoint (one descriptor byte each); final arrays are level-major --------------------------------------
__global__ __launch_bounds__(256) void k_orb_describe(const OrbDev *rois, int nlevels, const OrbTables *T)
{
    const OrbDev &R = rois[blockIdx.z / nlevels];
    const int level = blockIdx.z % nlevels;
    const int n2 = R.n2[level];
    const int kidx = blockIdx.x * 8 + (threadIdx.x >> 5);
    int off = 0;
    for (int l = 0; l < level; l++) off += R.n2[l];
    if (blockIdx.x == 0 && threadIdx.x == 0 && level == nlevels - 1) R.counters[1] = min(off + n2, R.cap);   // keypoints of the ROI
    if (kidx >= n2) return;
    const int byte = threadIdx.x & 31;
    const int x = R.k2_xy[((size_t)level * R.cap2 + kidx) * 2], y = R.k2_xy[((size_t)level * R.cap2 + kidx) * 2 + 1];
    const float ang_deg = R.k2_angle[(size_t)level * R.cap2 + kidx];
    const float sf = R.lscale[level];
    // computeKeyPoints tail: pt *= scale; computeOrbDescriptors: center = (cvRound(pt.y * (1.f/scale)), cvRound(pt.x * (1.f/scale)))
    const float px = (float)x * sf, py = (float)y * sf;
    const float inv = 1.f / sf;
    const int cx = cv_round_f(px * inv), cy = cv_round_f(py * inv);
    const float angle = ang_deg * (float)(3.1415926535897932384626433832795 / 180.f);
    double sd, cd;
    det_sincos((double)angle, &sd, &cd);                       // the oracle evaluates the same explicit algorithm (detmath.h)
    const float a = (float)cd, b = (float)sd;
    const int w = R.lw[level];
    g_cu8 center = (g_cu8)R.bl[level] + (size_t)cy * w + cx;
    const int *pat = T->pattern + byte * 32;
    int val = 0;
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        const int p0x = pat[4 * bit], p0y = pat[4 * bit + 1], p1x = pat[4 * bit + 2], p1y = pat[4 * bit + 3];
        const float x0 = p0x * a - p0y * b, y0 = p0x * b + p0y * a;
        const float x1 = p1x * a - p1y * b, y1 = p1x * b + p1y * a;
        const int t0 = center[cv_round_f(y0) * w + cv_round_f(x0)];
        const int t1 = center[cv_round_f(y1) * w + cv_round_f(x1)];
        val |= (t0 < t1) << bit;
    }
    const int o = off + kidx;
    if (o < R.cap) {
        R.desc[(size_t)o * 32 + byte] = (uint8_t)val;
        if (byte == 0) {
            R.kps_xy[2 * o] = px; R.kps_xy[2 * o + 1] = py;
            vfsms_keypoint kp;
            kp.x = px; kp.y = py; kp.size = (float)T->patch_size * sf; kp.angle = ang_deg;
            kp.response = R.k2_resp[(size_t)level * R.cap2 + kidx]; kp.octave = level; kp.class_id = -1;
            R.kps_out[o] = kp;
        }
    } else if (byte == 0) R.counters[2] = 1;
}

// ---- Hamming 1-NN for a batch of jobs + votes (BFMatcher("BruteForce-Hamming").match, ImageUtility.py:297-302) ------------------------
// lanes own queries (8 dwords in VGPRs), trains stream through the scalar path like the L2 matcher
typedef const uint32_t __attribute__((address_space(4))) cu32c;
// trains are split over blockIdx.z (ascending ranges); per-split first minima land in p_d1 / p_i1 and are merged in split order
__global__ __launch_bounds__(256) void k_bf_hamming_jobs(const MatchDev *jobs)
{
    const MatchDev &J = jobs[blockIdx.y];
    const int nq = __builtin_amdgcn_readfirstlane(*J.nq_ptr), nt = __builtin_amdgcn_readfirstlane(*J.nt_ptr);
    if ((int)(blockIdx.x * 256) >= nq) return;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const uint32_t *pq = reinterpret_cast<const uint32_t *>(J.q) + (size_t)min(max(q, 0), nq - 1) * 8;
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = pq[k];
    const int nsplit = gridDim.z, sp = blockIdx.z;
    const int chunk = (nt + nsplit - 1) / nsplit;
    const int t0 = sp * chunk, t1 = min(nt, t0 + chunk);
    int best = 0x7fffffff, bi = -1;
    cu32c *T = (cu32c *)(uintptr_t)J.t;
    for (int j = t0; j < t1; j++) {
        cu32c *tr = T + (size_t)j * 8;
        int d = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) d += __popc(v[k] ^ tr[k]);
        if (d < best) { best = d; bi = j; }              // first minimum wins
    }
    if (q >= nq) return;
    J.p_d1[(size_t)sp * J.capq + q] = (float)best; J.p_i1[(size_t)sp * J.capq + q] = bi;
}

__global__ __launch_bounds__(256) void k_hamming_merge(const MatchDev *jobs, int nsplit, int max_dist)
{
    const MatchDev &J = jobs[blockIdx.y];
    const int nq = *J.nq_ptr;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    int best = 0x7fffffff, bi = -1;
    for (int s = 0; s < nsplit; s++) {
        const int i = J.p_i1[(size_t)s * J.capq + q];
        const int d = i >= 0 ? (int)J.p_d1[(size_t)s * J.capq + q] : 0x7fffffff;
        if (d < best) { best = d; bi = i; }              // splits are ascending train ranges: the first minimum still wins
    }
    J.i1[q] = bi; J.d1[q] = (float)best; J.d2[q] = 0.f;
    int ok = bi >= 0 && (max_dist < 0 || best < max_dist);
    int vote = 0;
    if (ok && J.kq) {
        float ay = J.kq[2 * q + 1], ax = J.kq[2 * q];
        float by = J.kt[2 * bi + 1], bx = J.kt[2 * bi];
        int dx = (int)(ay - by), dy = (int)(ax - bx);
        vote = !(dx == 0 && dy == 0);
        J.votes[2 * (size_t)(J.capq + q)] = dx;
        J.votes[2 * (size_t)(J.capq + q) + 1] = dy;
    }
    J.match_flag[q] = ok | (vote << 1);
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------------
static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

int orb_level_dims(const vfsms_orb_params *p, int h, int w, int *lw, int *lh, float *lscale)
{
    for (int l = 0; l < p->n_levels; l++) {
        const float scale = (float)pow((double)p->scale_factor, (double)(l - p->first_level));
        lscale[l] = scale;
        lw[l] = (int)lrintf(w / scale); lh[l] = (int)lrintf(h / scale);
    }
    return VFSMS_OK;
}

size_t orb_roi_bytes(const vfsms_orb_params *p, int h, int w, int cap1, int cap2, int cap)
{
    int lw[VFSMS_ORB_MAX_LEVELS], lh[VFSMS_ORB_MAX_LEVELS]; float ls[VFSMS_ORB_MAX_LEVELS];
    orb_level_dims(p, h, w, lw, lh, ls);
    size_t b = 0;
    for (int l = 0; l < p->n_levels; l++) b += 4 * al((size_t)(lw[l] > 0 ? lw[l] : 1) * (lh[l] > 0 ? lh[l] : 1));
    b += al(sizeof(int) * 256 * p->n_levels) + al(256);
    b += p->n_levels * (al(sizeof(int) * 2 * cap1) + al(sizeof(float) * cap1) + al(sizeof(int) * 2 * cap2) + 2 * al(sizeof(float) * cap2));
    b += al(sizeof(float) * 2 * cap) + al((size_t)32 * cap) + al(sizeof(vfsms_keypoint) * cap);
    return b + 8192;
}

int orb_roi_carve(vfsms_ctx *ctx, OrbDev *r, const uint8_t *img, int stride, int h, int w, const vfsms_orb_params *p,
                  int cap1, int cap2, int cap)
{
    memset(r, 0, sizeof(*r));
    r->h = h; r->w = w; r->cap1 = cap1; r->cap2 = cap2; r->cap = cap;
    orb_level_dims(p, h, w, r->lw, r->lh, r->lscale);
    for (int l = 0; l < p->n_levels; l++) {
        const size_t n = (size_t)(r->lw[l] > 0 ? r->lw[l] : 1) * (r->lh[l] > 0 ? r->lh[l] : 1);
        if (l == p->first_level) { r->lv[l] = const_cast<uint8_t *>(img); r->ls[l] = stride; }
        else { r->lv[l] = (uint8_t *)ctx_arena_alloc(ctx, n); r->ls[l] = r->lw[l]; }
        r->bl[l] = (uint8_t *)ctx_arena_alloc(ctx, n);
        r->score[l] = (uint8_t *)ctx_arena_alloc(ctx, n);
        r->nms[l] = (uint8_t *)ctx_arena_alloc(ctx, n);
    }
    r->hist = (int *)ctx_arena_alloc(ctx, sizeof(int) * 256 * p->n_levels);
    r->counters = (int *)ctx_arena_alloc(ctx, 64 * sizeof(int));
    r->thr1 = r->counters + 16; r->n1 = r->counters + 32; r->n2 = r->counters + 48;
    r->k1_xy = (int *)ctx_arena_alloc(ctx, sizeof(int) * 2 * (size_t)cap1 * p->n_levels);
    r->k1_resp = (float *)ctx_arena_alloc(ctx, sizeof(float) * (size_t)cap1 * p->n_levels);
    r->k2_xy = (int *)ctx_arena_alloc(ctx, sizeof(int) * 2 * (size_t)cap2 * p->n_levels);
    r->k2_resp = (float *)ctx_arena_alloc(ctx, sizeof(float) * (size_t)cap2 * p->n_levels);
    r->k2_angle = (float *)ctx_arena_alloc(ctx, sizeof(float) * (size_t)cap2 * p->n_levels);
    r->kps_xy = (float *)ctx_arena_alloc(ctx, sizeof(float) * 2 * cap);
    r->desc = (uint8_t *)ctx_arena_alloc(ctx, (size_t)32 * cap);
    r->kps_out = (vfsms_keypoint *)ctx_arena_alloc(ctx, sizeof(vfsms_keypoint) * cap);
    if (!r->kps_out) { vfsms_set_error("arena exhausted while carving an ORB ROI"); return VFSMS_ERR_CAPACITY; }
    return VFSMS_OK;
}

int launch_orb(vfsms_ctx *ctx, const OrbDev *d_rois, const OrbDev *h_rois, int nrois, const vfsms_orb_params *p)
{
    if (nrois <= 0) return VFSMS_OK;
    const int nl = p->n_levels;
    int maxw = 0, maxh = 0, maxcap2 = 0;
    for (int r = 0; r < nrois; r++) {
        maxw = h_rois[r].w > maxw ? h_rois[r].w : maxw; maxh = h_rois[r].h > maxh ? h_rois[r].h : maxh;
        maxcap2 = h_rois[r].cap2 > maxcap2 ? h_rois[r].cap2 : maxcap2;
    }
    int maxcap1 = 0;
    for (int r = 0; r < nrois; r++) maxcap1 = h_rois[r].cap1 > maxcap1 ? h_rois[r].cap1 : maxcap1;
    // Runs of consecutive ROIs of one shape: the kernels whose grid is cut from the image size are launched once per run (a batch of the
    // incremental search mixes 409 x 2048 and 2048 x 409 strips; a grid for the largest height AND width dispatched five times the
    // workgroups either shape needs).  vfsms_attempt_orb_batch orders its ROIs by shape.
    struct Run { int first, count, h, w; };
    std::vector<Run> runs;
    for (int r = 0; r < nrois; r++) {
        if (!runs.empty() && runs.back().h == h_rois[r].h && runs.back().w == h_rois[r].w) runs.back().count++;
        else { Run q; q.first = r; q.count = 1; q.h = h_rois[r].h; q.w = h_rois[r].w; runs.push_back(q); }
    }
    (void)maxw; (void)maxh;
    hipLaunchKernelGGL(k_orb_clear, dim3(nrois), dim3(256), 0, ctx->stream, d_rois, nl);
    {
        ProfScope ps(ctx, "orb_pyramid");
        for (int l = 1; l < nl; l++)
            for (const Run &q : runs) {
                int lw[VFSMS_ORB_MAX_LEVELS], lh[VFSMS_ORB_MAX_LEVELS]; float ls[VFSMS_ORB_MAX_LEVELS];
                orb_level_dims(p, q.h, q.w, lw, lh, ls);
                if (lw[l] <= 0 || lh[l] <= 0) continue;
                hipLaunchKernelGGL(k_orb_resize, dim3((lw[l] + 256) / 256, lh[l] + 1, q.count), dim3(256), 0, ctx->stream, d_rois + q.first, l);
            }
    }
    {
        ProfScope ps(ctx, "orb_fast");
        for (const Run &q : runs) {
            int lw[VFSMS_ORB_MAX_LEVELS], lh[VFSMS_ORB_MAX_LEVELS]; float ls[VFSMS_ORB_MAX_LEVELS];
            orb_level_dims(p, q.h, q.w, lw, lh, ls);
            OrbPlan plan; plan.first[0] = 0;
            for (int l = 0; l < nl; l++) {
                const int tx = std::max((lw[l] + FT_W - 1) / FT_W, 1), ty = std::max((lh[l] + FT_H - 1) / FT_H, 0);
                plan.tiles_x[l] = tx; plan.first[l + 1] = plan.first[l] + tx * ty;
            }
            if (plan.first[nl] > 0)
                hipLaunchKernelGGL(k_orb_fast_nms, dim3(plan.first[nl], q.count), dim3(256), 0, ctx->stream, d_rois + q.first, nl,
                                   p->fast_threshold < 0 ? 0 : p->fast_threshold > 255 ? 255 : p->fast_threshold, p->edge_threshold, plan);
        }
        hipLaunchKernelGGL(k_orb_threshold, dim3((nrois * nl + 63) / 64), dim3(64), 0, ctx->stream, d_rois, nrois, nl, ctx->d_orb_tables);
    }
    {
        ProfScope ps(ctx, "orb_select");
        hipLaunchKernelGGL(k_orb_count, dim3(ORB_CHUNKS, nrois * nl), dim3(1024), 0, ctx->stream, d_rois, nl);
        hipLaunchKernelGGL(k_orb_chunk_scan, dim3(nrois * nl), dim3(256), 0, ctx->stream, d_rois, nl);
        hipLaunchKernelGGL(k_orb_scatter, dim3(ORB_CHUNKS, nrois * nl), dim3(1024), 0, ctx->stream, d_rois, nl);
        hipLaunchKernelGGL(k_orb_harris, dim3((maxcap1 + 255) / 256, nrois * nl), dim3(256), 0, ctx->stream, d_rois, nl);
        hipLaunchKernelGGL(k_orb_select2, dim3(nrois * nl), dim3(1024), 0, ctx->stream, d_rois, nl, ctx->d_orb_tables);
        hipLaunchKernelGGL(k_orb_angle, dim3((maxcap2 + 3) / 4, nrois * nl), dim3(256), 0, ctx->stream, d_rois, nl, ctx->d_orb_tables);
    }
    {
        ProfScope ps(ctx, "orb_describe");
        for (const Run &q : runs) {
            int lw[VFSMS_ORB_MAX_LEVELS], lh[VFSMS_ORB_MAX_LEVELS]; float ls[VFSMS_ORB_MAX_LEVELS];
            orb_level_dims(p, q.h, q.w, lw, lh, ls);
            OrbPlan plan; plan.first[0] = 0;
            for (int l = 0; l < nl; l++) {
                const int tx = std::max((lw[l] + 31) / 32, 1), ty = std::max((lh[l] + 31) / 32, 0);
                plan.tiles_x[l] = tx; plan.first[l + 1] = plan.first[l] + tx * ty;
            }
            if (plan.first[nl] > 0)
                hipLaunchKernelGGL(k_orb_blur, dim3(plan.first[nl], q.count), dim3(256), 0, ctx->stream, d_rois + q.first, nl, ctx->d_orb_tables, plan);
        }
        hipLaunchKernelGGL(k_orb_describe, dim3((maxcap2 + 7) / 8, 1, nrois * nl), dim3(256), 0, ctx->stream, d_rois, nl, ctx->d_orb_tables);
    }
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

int launch_hamming_mode(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, int nsplit, int max_dist, int offset_evaluate)
{
    if (njobs <= 0) return VFSMS_OK;
    {
        ProfScope ps(ctx, "bf_hamming");
        hipLaunchKernelGGL(k_bf_hamming_jobs, dim3((capq + 255) / 256, njobs, nsplit), dim3(256), 0, ctx->stream, d_jobs);
        hipLaunchKernelGGL(k_hamming_merge, dim3((capq + 255) / 256, njobs), dim3(256), 0, ctx->stream, d_jobs, nsplit, max_dist);
    }
    HIP_TRY(hipGetLastError());
    return launch_scan_mode(ctx, d_jobs, njobs, capq, offset_evaluate);
}

// ---- tables: per-level quotas, umax, fixed-point Gaussian, sampling pattern (uploaded when the parameters change) --------------------
int ctx_prepare_orb(vfsms_ctx *ctx, const vfsms_orb_params *p)
{
    if (!p || p->n_levels < 1 || p->n_levels > VFSMS_ORB_MAX_LEVELS || p->n_features < 0 || p->scale_factor <= 1.f ||
        p->patch_size < 2 || p->edge_threshold < 0) { vfsms_set_error("bad ORB parameters"); return VFSMS_ERR_BAD_ARG; }
    if (p->first_level != 0 || p->wta_k != 2 || p->score_type != 0 || p->patch_size > 31 || p->edge_threshold < p->patch_size / 2 + 1) {
        vfsms_set_error("ORB: only first_level 0, WTA_K 2, HARRIS score, patch_size <= 31, edge_threshold > patch_size/2 are supported");
        return VFSMS_ERR_UNSUPPORTED;
    }
    if (ctx->orb_valid && memcmp(&ctx->cur_orb, p, sizeof(*p)) == 0) return VFSMS_OK;
    OrbTables T;
    memset(&T, 0, sizeof(T));
    const double scaleFactor = (double)p->scale_factor;
    {   // computeKeyPoints: nfeaturesPerLevel
        const float factor = (float)(1.0 / scaleFactor);
        float nd = p->n_features * (1 - factor) / (1 - (float)pow((double)factor, (double)p->n_levels));
        int sum = 0;
        for (int l = 0; l < p->n_levels - 1; l++) { T.nfeat[l] = (int)lrintf(nd); sum += T.nfeat[l]; nd *= factor; }
        T.nfeat[p->n_levels - 1] = p->n_features - sum > 0 ? p->n_features - sum : 0;
    }
    {   // umax: end of each row of the circular patch
        const int hp = p->patch_size / 2;
        T.half_patch = hp; T.patch_size = p->patch_size;
        int v, v0;
        const int vmax = (int)floor(hp * sqrtf(2.f) / 2 + 1), vmin = (int)ceil(hp * sqrtf(2.f) / 2);
        for (v = 0; v <= vmax; ++v) T.umax[v] = (int)lrint(sqrt((double)hp * hp - v * v));
        for (v = hp, v0 = 0; v >= vmin; --v) {
            while (T.umax[v0] == T.umax[v0 + 1]) ++v0;
            T.umax[v] = v0;
            ++v0;
        }
    }
    {   // getGaussianKernel(7, 2, CV_32F) -> 8-bit fixed point
        float cf[7]; double sum = 0; const double s2 = -0.5 / (2.0 * 2.0);
        for (int i = 0; i < 7; i++) { const double x = i - 3.0; cf[i] = (float)exp(s2 * x * x); sum += cf[i]; }
        sum = 1. / sum;
        for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); T.kf[i] = (int)lrintf(cf[i] * 256.f); }
    }
    if (p->patch_size == 31) {
        // upstream's learned table bit_pattern_31_ (what cv2.ORB_create(..., patchSize = 31, ...) samples, ImageUtility.py:260)
        for (int i = 0; i < 1024; i++) T.pattern[i] = VFSMS_ORB_BIT_PATTERN_31[i];
    } else {   // makeRandomPattern: RNG(0x34985739), MWC -- upstream's generator for every other patch size
        uint64_t state = 0x34985739ULL;
        const int a = -p->patch_size / 2, b = p->patch_size / 2 + 1;
        for (int i = 0; i < 1024; i++) {
            state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
            T.pattern[i] = (int)((unsigned)state % (unsigned)(b - a) + a);
        }
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (!ctx->d_orb_tables) HIP_TRY(hipMalloc((void **)&ctx->d_orb_tables, sizeof(OrbTables)));
    HIP_TRY(hipMemcpy(ctx->d_orb_tables, &T, sizeof(T), hipMemcpyHostToDevice));
    ctx->cur_orb = *p; ctx->orb_valid = true;
    return VFSMS_OK;
}

// surf_kernels.hip -- integral image, fast-Hessian pyramid, NMS + sub-pixel interpolation, keypoint
// ordering, orientation and descriptor kernels for gfx950 (wave64).
//
// What is computed follows the reference's SURF call (cv2.xfeatures2d.SURF_create().detectAndCompute,
// ImageUtility.py:258,262; GPU twin appendix/myGpuFeatures.cpp:67-104) -- i.e. OpenCV 3.3.1 xfeatures2d
// semantics as written down in SURVEY.md section 8a D1-D5 / Appendix A.2.  How it is computed is native:
// every kernel is batched over an array of ROI working sets (blockIdx.z / .y = ROI), no host sync between
// stages (grids are sized by capacity, blocks early-exit on device-side counts), all float/double
// operation orders are explicit (built with -ffp-contract=off) so results are reproducible bit for bit.
#include "common.h"
#include "detmath.h"
#include <math.h>
#include <float.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

// ---------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------
// Pointers that travel through the RoiDev records in memory are generic ("flat") to the compiler: it would emit
// flat_load/flat_store, which tick BOTH vmcnt and lgkmcnt (so they serialise against LDS traffic) and take the
// aperture-check path.  Everything in a RoiDev is device global memory, so say so.
#define GAS __attribute__((address_space(1)))
typedef GAS const uint8_t *g_cu8;
typedef GAS const int32_t *g_ci32;
typedef GAS float *g_f32;
typedef GAS const float *g_cf32;
typedef GAS int32_t *g_i32;
__device__ __forceinline__ int cv_round_f(float v) { return (int)rintf(v); }      // round-half-even
// (uchar)cvRound(v) for 0 <= v < 2^22 in ONE VOP2 add: v + 1.5 * 2^23 has an ulp of 1, so the sum is v rounded half-to-even and the integer
// sits in the low mantissa bits -- the byte store takes it from there (v_rndne_f32 + v_cvt_i32_f32 before).
__device__ __forceinline__ uint8_t round_u8_pos(float v) { return (uint8_t)__float_as_uint(v + 12582912.f); }
__device__ __forceinline__ int cv_round_d(double v) { return (int)rint(v); }
__device__ __forceinline__ int cv_floor_d(double v) { return (int)floor(v); }
__device__ __forceinline__ int cv_ceil_d(double v) { return (int)ceil(v); }

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

// ---------------------------------------------------------------------------------------------------
// K1 integral image: u8 h x w -> i32 (h+1) x (w+1)            [HBM-bound; algorithmic bytes = h*w + 4(h+1)(w+1)]
//   Every pixel is read twice as a byte and every output written ONCE as an int (the previous two-pass form read and
//   re-wrote the int32 plane: 25 B/px of traffic).  The ROI is cut into bands of INT_TH rows:
//     k_integral_bandsum : per band, the sum of every column over the band's rows        (reads 1 B/px)
//     k_integral_bandscan: exclusive scan of those column sums over the bands             (4/INT_TH B/px each way)
//     k_integral_final   : a workgroup owns one band x all columns.  A thread owns 4 adjacent columns: it keeps the band's
//                          16 pixel dwords in registers, adds them onto the band carry (running column sums), and the row
//                          prefix over the 4-column groups is a wave shuffle scan + one LDS exchange for all 16 rows at once
//                          (one barrier per 1024-column chunk).  Each lane stores one int4 per row: a wave writes 1 KB of
//                          contiguous output per instruction.                             (reads 1 B/px, writes 4 B/px)
// ---------------------------------------------------------------------------------------------------
#define INT_TH 16
typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32u __attribute__((aligned(1)));

// 4 adjacent pixels of a row as one dword (unaligned global loads are legal on gfx950).  At the ragged right edge the load is
// moved left so that it stays inside the row and the bytes are shifted down: columns >= w read as zero.
__device__ __forceinline__ uint32_t load_px4(g_cu8 row, int x, int w)      // requires w >= 4
{
    const int xa = min(x, w - 4);
    const uint32_t v = *(GAS const u32u *)(row + xa);
    return v >> (8 * (x - xa));                             // x - xa is 0 except for the last group of a row (1..3)
}

__global__ __launch_bounds__(256) void k_integral_bandsum(const RoiDev *rois)
{
    const RoiDev &R = rois[blockIdx.z];
    const int h = R.h, w = R.w, stride = R.stride, ipitch = R.ipitch;
    const int y0 = blockIdx.y * INT_TH;
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (y0 >= h || x >= w || w < 4) return;
    const int rows = min(INT_TH, h - y0);
    g_cu8 src = (g_cu8)R.img + (size_t)y0 * stride;
    g_i32 dst = (g_i32)R.icarry + (size_t)blockIdx.y * ipitch + x;
    uint32_t px[INT_TH];
#pragma unroll
    for (int r = 0; r < INT_TH; r++) px[r] = r < rows ? load_px4(src + (size_t)r * stride, x, w) : 0u;
    uint32_t even = 0, odd = 0;                       // 16-bit lanes: 16 rows x 255 < 65536
#pragma unroll
    for (int r = 0; r < INT_TH; r++) { even += px[r] & 0x00ff00ffu; odd += (px[r] >> 8) & 0x00ff00ffu; }
    int4u o = {(int)(even & 0xffff), (int)(odd & 0xffff), (int)(even >> 16), (int)(odd >> 16)};
    *(GAS int4u *)dst = o;
}

__global__ __launch_bounds__(256) void k_integral_bandscan(const RoiDev *rois)
{
    const RoiDev &R = rois[blockIdx.y];
    const int ipitch = R.ipitch;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= ipitch) return;
    const int nb = (R.h + INT_TH - 1) / INT_TH;
    g_i32 c = (g_i32)R.icarry + x;
    int run = 0;
    // 16 bands per trip: the loads of a trip are all issued before its stores (in-place update: the compiler must otherwise order
    // every load behind the previous store, one memory round trip per band)
    for (int b0 = 0; b0 < nb; b0 += 16) {
        int v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = b0 + k < nb ? c[(size_t)(b0 + k) * ipitch] : 0;
#pragma unroll
        for (int k = 0; k < 16; k++) { if (b0 + k < nb) c[(size_t)(b0 + k) * ipitch] = run; run += v[k]; }
    }
}

__global__ __launch_bounds__(256) void k_integral_final(const RoiDev *rois)
{
    const RoiDev &R = rois[blockIdx.y];
    // every field into a local first: the compiler cannot prove that the stores below do not alias the ROI record
    const int h = R.h, w = R.w, stride = R.stride, ipitch = R.ipitch;
    g_cu8 img = (g_cu8)R.img;
    g_i32 S = (g_i32)R.sum;
    g_ci32 carry = (g_ci32)R.icarry + (size_t)blockIdx.x * ipitch;
    const int y0 = blockIdx.x * INT_TH;
    if (y0 >= h) return;
    const int rows = min(INT_TH, h - y0);
    const int sw = w + 1;
    if (w < 4) {                                                                        // degenerate strips: one thread, plain loops
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            for (int xx = 0; xx < sw; xx++) S[xx] = 0;
            for (int y = 0; y < h; y++) {
                int rs = 0;
                S[(size_t)(y + 1) * sw] = 0;
                for (int xx = 0; xx < w; xx++) { rs += img[(size_t)y * stride + xx]; S[(size_t)(y + 1) * sw + xx + 1] = S[(size_t)y * sw + xx + 1] + rs; }
            }
        }
        return;
    }
    if (blockIdx.x == 0)
        for (int xx = threadIdx.x; xx < sw; xx += 256) S[xx] = 0;                       // row 0 of the integral is zero
    if (threadIdx.x < rows) S[(size_t)(y0 + 1 + threadIdx.x) * sw] = 0;                // column 0 too
    __shared__ int wsum[INT_TH][4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    g_cu8 src = img + (size_t)y0 * stride;
    int ccarry[INT_TH];                                                                 // row sums of the chunks to the left
#pragma unroll
    for (int r = 0; r < INT_TH; r++) ccarry[r] = 0;
    for (int c0 = 0; c0 < w; c0 += 1024) {
        const int x = c0 + threadIdx.x * 4;
        const bool in = x < w;
        const int xl = in ? x : 0;
        uint32_t px[INT_TH];
#pragma unroll
        for (int r = 0; r < INT_TH; r++) {
            const uint32_t v = load_px4(src + (size_t)min(r, rows - 1) * stride, xl, w);   // always a legal address
            px[r] = (in && r < rows) ? v : 0u;
        }
        int4u C = *(GAS const int4u *)(carry + xl);
        if (!in) { C.x = 0; C.y = 0; C.z = 0; C.w = 0; }
        // phase 1: total of this thread's 4 columns in every row (band carry + running pixel sums), scanned across the workgroup
        int tot[INT_TH];
        int run = C.x + C.y + C.z + C.w;
#pragma unroll
        for (int r = 0; r < INT_TH; r++) {
            run += (int)__builtin_amdgcn_sad_u8(px[r], 0u, 0u);                          // sum of the 4 bytes
            tot[r] = run;
        }
        int incl[INT_TH];
#pragma unroll
        for (int r = 0; r < INT_TH; r++) incl[r] = tot[r];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int r = 0; r < INT_TH; r++) {
                const int n = __shfl_up(incl[r], d, 64);
                incl[r] += lane >= d ? n : 0;
            }
        }
        if (c0) __syncthreads();                                                         // previous chunk's wsum fully consumed
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < INT_TH; r++) wsum[r][wid] = incl[r];
        }
        __syncthreads();
        // phase 2: running column sums again, now with the exclusive row prefix in front
        int t0 = C.x, t1 = C.y, t2 = C.z, t3 = C.w;
        g_i32 dst = S + (size_t)(y0 + 1) * sw + 1 + xl;
        const bool whole = x + 3 < w;
#pragma unroll
        for (int r = 0; r < INT_TH; r++) {
            const int4 ws = *reinterpret_cast<const int4 *>(&wsum[r][0]);
            int off = ccarry[r] + incl[r] - tot[r];
            off += wid > 0 ? ws.x : 0; off += wid > 1 ? ws.y : 0; off += wid > 2 ? ws.z : 0;
            ccarry[r] += ws.x + ws.y + ws.z + ws.w;
            t0 += (int)(px[r] & 0xff); t1 += (int)((px[r] >> 8) & 0xff); t2 += (int)((px[r] >> 16) & 0xff); t3 += (int)(px[r] >> 24);
            if (r < rows) {
                int4u o;
                o.x = off + t0; o.y = o.x + t1; o.z = o.y + t2; o.w = o.z + t3;
                if (whole) *(GAS int4u *)(dst + (size_t)r * sw) = o;
                else if (in) { dst[(size_t)r * sw] = o.x; if (x + 1 < w) dst[(size_t)r * sw + 1] = o.y; if (x + 2 < w) dst[(size_t)r * sw + 2] = o.z; }
            }
        }
    }
}

// Kernels whose grid is cut from the image size are launched once per RUN of consecutive ROIs of one shape: a batch of the incremental search
// mixes the 409 x 2048 strips of the column pairs with the 2048 x 409 strips of the turn candidates, and a grid sized for the largest height
// AND the largest width of the batch dispatched five times the workgroups either shape needs (empty ones exit at once, but a batch of 96 ROIs
// paid 1.3 ms per launch for dispatching them).  Callers order their ROIs by shape (attempt_surf_impl).
struct ShapeRun { int first, count, h, w; };
static std::vector<ShapeRun> shape_runs(const RoiDev *h_rois, int nrois)
{
    std::vector<ShapeRun> runs;
    for (int r = 0; r < nrois; r++) {
        if (!runs.empty() && runs.back().h == h_rois[r].h && runs.back().w == h_rois[r].w) runs.back().count++;
        else { ShapeRun q; q.first = r; q.count = 1; q.h = h_rois[r].h; q.w = h_rois[r].w; runs.push_back(q); }
    }
    return runs;
}

int launch_integral(vfsms_ctx *ctx, const RoiDev *d_rois, int nrois, int maxh, int maxw)
{
    if (nrois <= 0) return VFSMS_OK;
    const int nb = (maxh + INT_TH - 1) / INT_TH;
    hipLaunchKernelGGL(k_integral_bandsum, dim3((maxw + 1023) / 1024, nb, nrois), dim3(256), 0, ctx->stream, d_rois);
    hipLaunchKernelGGL(k_integral_bandscan, dim3((maxw + 3 + 255) / 256, nrois), dim3(256), 0, ctx->stream, d_rois);
    hipLaunchKernelGGL(k_integral_final, dim3(nb, nrois), dim3(256), 0, ctx->stream, d_rois);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}
size_t integral_carry_bytes(int h, int w) { return sizeof(int32_t) * (size_t)((h + INT_TH - 1) / INT_TH) * (size_t)((w + 3) & ~3); }

// ---------------------------------------------------------------------------------------------------
// K2 fast-Hessian determinant, one launch per octave, blockIdx.z = roi * (nLayers+2) + layer
//   calcLayerDetAndTrace: box sums are int, each multiplied by its float weight as float, accumulated
//   in double, cast to float; det = dx*dy - 0.81f*dxy*dxy.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float haar_box(g_ci32 sp, int sw, const LayerPat &P, int k0, int n)
{
    double d = 0;
    for (int k = k0; k < k0 + n; k++) {
        const int dx1 = P.box[k][0], dy1 = P.box[k][1], dx2 = P.box[k][2], dy2 = P.box[k][3];
        int v = sp[dy1 * sw + dx1] + sp[dy2 * sw + dx2] - sp[dy2 * sw + dx1] - sp[dy1 * sw + dx2];
        d += (double)((float)v * P.w[k]);
    }
    return (float)d;
}

// KeyPoint::class_id = sign(trace) of the candidate's own sample, trace = dx + dy of calcLayerDetAndTrace recomputed from the integral image
// for the few thousand candidates of an ROI (one dense thread each, in the ordering kernels) -- the trace LAYERS are not materialised:
// round 2 wrote 4 B per sample and layer (277 MB per launch) for the sign of a fraction of a per cent of the cells.
__device__ __forceinline__ int cand_class_id(const RoiDev &R, const LayerPat *pats, const Cand &c)
{
    const LayerPat &P = pats[c.layer];
    const int ss = P.step, size = P.size;
    const int sum_i = ss * (c.i - (size / 2) / ss), sum_j = ss * (c.j - (size / 2) / ss);
    g_ci32 sp = (g_ci32)R.sum + (size_t)sum_i * (R.w + 1) + sum_j;
    const float tr = haar_box(sp, R.w + 1, P, 0, 3) + haar_box(sp, R.w + 1, P, 3, 3);
    return (tr > 0) - (tr < 0);
}

// Gather variant (coarse octaves; every octave when n_octave_layers != 3): ONE launch for all its octaves -- first[q] counts the
// 64 x 4-sample tiles of the octaves before o0 + q -- so the small octaves share a launch and fill each other's tails.
struct HessPlan { int o0, noct; int first[VFSMS_MAX_OCTAVES + 1]; int tiles_x[VFSMS_MAX_OCTAVES]; };

__global__ __launch_bounds__(256) void k_hessian(const RoiDev *rois, const LayerPat *pats, int layers_per_octave, HessPlan plan, int nrois)
{
    unsigned roi, inner;
    xcd_roi_map(blockIdx.x, (unsigned)(plan.first[plan.noct] * layers_per_octave), (unsigned)nrois, roi, inner);
    const int l = (int)(inner % (unsigned)layers_per_octave);
    int t = (int)(inner / (unsigned)layers_per_octave);
    int q = 0;
    while (q + 1 < plan.noct && t >= plan.first[q + 1]) q++;
    t -= plan.first[q];
    const int li = (plan.o0 + q) * layers_per_octave + l;
    const RoiDev &R = rois[roi];
    const LayerPat &P = pats[li];
    const int step = P.step, size = P.size;
    if (size > R.h || size > R.w) return;
    const int samples_i = 1 + (R.h - size) / step;
    const int samples_j = 1 + (R.w - size) / step;
    const int j = (t % plan.tiles_x[q]) * 64 + threadIdx.x;
    const int i = (t / plan.tiles_x[q]) * 4 + threadIdx.y;
    if (i >= samples_i || j >= samples_j) return;
    const int sw = R.w + 1;
    const int lcols = R.w / step;
    g_ci32 sp = (g_ci32)R.sum + (size_t)(i * step) * sw + j * step;
    float dx = haar_box(sp, sw, P, 0, 3);
    float dy = haar_box(sp, sw, P, 3, 3);
    float dxy = haar_box(sp, sw, P, 6, 4);
    size_t o = (size_t)(i + P.margin) * lcols + (j + P.margin);
    ((g_f32)R.det[li])[o] = dx * dy - 0.81f * dxy * dxy;     // (the trace is recomputed for the candidates: cand_class_id)
}

// ---- the determinant of one sample from its 32 distinct integral taps (round 6; shared by the LDS, row-staged and gather kernels) --------------
// A tap is named by the PATTERN coordinates (vy, vx), 0..9, of the 9 x 9 box filters; its offset in a layer of `SIZE` px is
// hcorner<SIZE>(v) = cvRound(SIZE / 9.f * v) (vfsms_haar_corner).  What changed against the box-by-box form (3 integer operations per
// box, 30 per sample): the three Dx boxes share their two rows and the three Dy boxes their two columns, so the differences across the
// pair of rows (columns) are formed once -- V(c) = S(r7, c) - S(r2, c) for the four Dx columns, then box k = V(c_k+1) - V(c_k): 7
// operations instead of 9, two's-complement arithmetic is associative, so every box sum is the same int -- and the first addend of a
// wavelet is not added to 0.0 any more: its weight is positive and box sums of an integral image of bytes are non-negative, so the
// product is never -0 and 0.0 + p == p bit for bit.  Float products and their double accumulation order are calcHaarPattern's.
template <int SIZE> constexpr int hcorner(int v) { return (2 * SIZE * v + 9) / 18; }
template <int SIZE, class Tap>
__device__ __forceinline__ float hess_det(const Tap &T, const float (&w)[10])
{
    // Dx: boxes {0,2,3,7}, {3,2,6,7}, {6,2,9,7} as (x1, y1, x2, y2): rows 2 and 7, columns 0, 3, 6, 9
    const int vx0 = T.template at<7, 0>() - T.template at<2, 0>(), vx3 = T.template at<7, 3>() - T.template at<2, 3>();
    const int vx6 = T.template at<7, 6>() - T.template at<2, 6>(), vx9 = T.template at<7, 9>() - T.template at<2, 9>();
    double d = (double)((float)(vx3 - vx0) * w[0]);
    d += (double)((float)(vx6 - vx3) * w[1]);
    d += (double)((float)(vx9 - vx6) * w[2]);
    const float dx = (float)d;
    // Dy: boxes {2,0,7,3}, {2,3,7,6}, {2,6,7,9}: columns 2 and 7, rows 0, 3, 6, 9
    const int hy0 = T.template at<0, 7>() - T.template at<0, 2>(), hy3 = T.template at<3, 7>() - T.template at<3, 2>();
    const int hy6 = T.template at<6, 7>() - T.template at<6, 2>(), hy9 = T.template at<9, 7>() - T.template at<9, 2>();
    d = (double)((float)(hy3 - hy0) * w[3]);
    d += (double)((float)(hy6 - hy3) * w[4]);
    d += (double)((float)(hy9 - hy6) * w[5]);
    const float dy = (float)d;
    // Dxy: boxes {1,1,4,4}, {5,1,8,4}, {1,5,4,8}, {5,5,8,8}
    const int b6 = T.template at<1, 1>() + T.template at<4, 4>() - T.template at<4, 1>() - T.template at<1, 4>();
    const int b7 = T.template at<1, 5>() + T.template at<4, 8>() - T.template at<4, 5>() - T.template at<1, 8>();
    const int b8 = T.template at<5, 1>() + T.template at<8, 4>() - T.template at<8, 1>() - T.template at<5, 4>();
    const int b9 = T.template at<5, 5>() + T.template at<8, 8>() - T.template at<8, 5>() - T.template at<5, 8>();
    d = (double)((float)b6 * w[6]);
    d += (double)((float)b7 * w[7]);
    d += (double)((float)b8 * w[8]);
    d += (double)((float)b9 * w[9]);
    const float dxy = (float)d;
    return dx * dy - 0.81f * dxy * dxy;
}

// LDS-tiled variant for the fine octaves (0 and 1 hold 94 % of the samples): the five layers of an octave read the
// same integral-image neighbourhood, so one workgroup stages the (tile + largest wavelet) window of the integral once
// and evaluates all 5 x 40 taps of its TW x 16 samples from LDS -- 200 L2 gathers per sample become ~5 coalesced loads.
// Arithmetic and its order are those of k_hessian.
// The box corners of a layer depend on its size alone ((9 + 6 l) << octave; vfsms_haar_corner), so with the five layers and ten boxes
// unrolled every LDS tap is `ds_read_b32 base offset:imm` -- no address arithmetic per tap; the weights stay the host's floats.
template <int SIZE, int STEP, int LW, int LWH, int PLANE>
#ifndef HESS_O1_WAVES
#define HESS_O1_WAVES 8
#endif
struct HessLdsTap {
    const int32_t *sp;
    template <int VY, int VX> __device__ __forceinline__ int at() const
    {
        constexpr int dy = hcorner<SIZE>(VY), dx = hcorner<SIZE>(VX);
        return STEP == 2 ? sp[(dx & 1) * PLANE + dy * LWH + (dx >> 1)] : sp[dy * LW + dx];
    }
};
// one layer of a staged tile: TW x TH samples from the LDS window (k_hessian_lds)
template <int STEP, int TW, int TH, int LW, int LWH, int PLANE, int NT, int L>
__device__ __forceinline__ void hessian_lds_layer(const RoiDev &R, const LayerPat *pats, const int li, const int32_t *tile, const int i0, const int j0, const int tid)
{
    constexpr int OCT = STEP == 1 ? 0 : 1;
    constexpr int size = (9 + 6 * L) << OCT;                      // == P.size (checked by ctx_prepare_surf)
    if (size > R.h || size > R.w) return;
    const LayerPat &P = pats[li];
    const int samples_i = 1 + (R.h - size) / STEP, samples_j = 1 + (R.w - size) / STEP;
    g_f32 det = (g_f32)R.det[li];
    const int lcols = R.w / STEP;
    constexpr int margin = (size / 2) / STEP;
    float w[10];
#pragma unroll
    for (int k = 0; k < 10; k++) w[k] = P.w[k];
    for (int e = tid; e < TW * TH; e += NT) {
        const int ly = e / TW, lx = e - ly * TW;
        const int i = i0 + ly, j = j0 + lx;
        if (i >= samples_i || j >= samples_j) continue;
        const int32_t *sp = STEP == 2 ? tile + (ly * STEP) * LWH + lx : tile + (ly * STEP) * LW + lx * STEP;
        det[(size_t)(i + margin) * lcols + (j + margin)] = hess_det<size>(HessLdsTap<size, STEP, LW, LWH, PLANE>{sp}, w);
    }
}

// NW waves per workgroup (round 6: octave 1 runs EIGHT -- its 50 KB window admits three workgroups per CU whatever their size, and the kernel
// waited 57 % of its wave cycles with twelve waves per CU; 24 hide more of the staging and tap latency for the same LDS)
template <int STEP, int TW, int NW>
__global__ __launch_bounds__(NW * 64) void k_hessian_lds(const RoiDev *rois, const LayerPat *pats, int layers_per_octave, int octave)
{
    constexpr int TH = 16;
    constexpr int MAXSZ = 33 * STEP;                              // size of the octave's coarsest layer: (9 + 6*4) << o
    constexpr int LW = (TW - 1) * STEP + MAXSZ + 1, LH = (TH - 1) * STEP + MAXSZ + 1;
    // STEP 2: the lanes of a wave sample every second column, so a row-major window is read with a stride of two dwords -- two lanes per
    // LDS bank on every tap, and the taps are what bounds this kernel.  The window is therefore kept as two planes, even and odd columns:
    // column 2 lx + dx of a tap lies in plane dx & 1 at lx + (dx >> 1) -- unit stride across the lanes, both still immediates.
    constexpr int LWH = STEP == 2 ? (LW + 1) / 2 : LW;            // row pitch (of a plane)
    constexpr int PLANE = STEP == 2 ? LH * LWH + 1 : 0;           // (+1: the two planes start in different banks)
    __shared__ int32_t tile[STEP == 2 ? 2 * (LH * LWH + 1) : LH * LW];
    const RoiDev &R = rois[blockIdx.z];
    const int sw = R.w + 1, sh = R.h + 1;
    const int j0 = blockIdx.x * TW, i0 = blockIdx.y * TH;          // sample coordinates of the tile origin
    if (j0 * STEP >= R.w || i0 * STEP >= R.h) return;
    g_ci32 S = (g_ci32)R.sum;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    {
        // Round 6: a wave stages whole window rows (the row address is scalar), its lanes the columns lane, lane + 64, ...: two VALU
        // instructions per element (the load and the LDS store at a lane-constant offset) where the flat loop over LH * LW elements
        // spent ~20 on div / mod, clamps, 64-bit addresses and plane offsets -- half of octave 1's instructions, whose 129 x 97 window
        // feeds 512 samples.
        constexpr int NK = (LW + 63) / 64;
        const int lane = threadIdx.x, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
        const int x0 = j0 * STEP, y0 = i0 * STEP;
        uint32_t gxo[NK]; int lo[NK];
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int tx = lane + 64 * k;
            gxo[k] = (uint32_t)min(x0 + tx, sw - 1);
            lo[k] = STEP == 2 ? (tx & 1) * PLANE + (tx >> 1) : tx;
        }
        constexpr int RB = 4;                                     // rows in flight per wave: RB * NK loads are issued before the first store
        for (int tb = wave; tb < LH; tb += NW * RB) {
            int32_t v[RB][NK];
#pragma unroll
            for (int r = 0; r < RB; r++) {
                g_ci32 row = S + (size_t)min(y0 + min(tb + NW * r, LH - 1), sh - 1) * sw;
#pragma unroll
                for (int k = 0; k < NK; k++) v[r][k] = row[gxo[k]];
            }
#pragma unroll
            for (int r = 0; r < RB; r++) {
                const int ty = tb + NW * r;
                if (ty < LH) {
                    int32_t *trow = tile + ty * LWH;
#pragma unroll
                    for (int k = 0; k < NK; k++)
                        if (NK * 64 == LW || k + 1 < NK || lane + 64 * k < LW) trow[lo[k]] = v[r][k];
                }
            }
        }
    }
    __syncthreads();
    const int lb = octave * layers_per_octave;
    hessian_lds_layer<STEP, TW, TH, LW, LWH, PLANE, NW * 64, 0>(R, pats, lb + 0, tile, i0, j0, tid);
    hessian_lds_layer<STEP, TW, TH, LW, LWH, PLANE, NW * 64, 1>(R, pats, lb + 1, tile, i0, j0, tid);
    hessian_lds_layer<STEP, TW, TH, LW, LWH, PLANE, NW * 64, 2>(R, pats, lb + 2, tile, i0, j0, tid);
    hessian_lds_layer<STEP, TW, TH, LW, LWH, PLANE, NW * 64, 3>(R, pats, lb + 3, tile, i0, j0, tid);
    hessian_lds_layer<STEP, TW, TH, LW, LWH, PLANE, NW * 64, 4>(R, pats, lb + 4, tile, i0, j0, tid);
}

// Coarse octaves (2, 3) with the stock five layers: the (tile + wavelet) window of an octave-2 tile would be 120-200 KB of LDS, so the taps
// stay gathers from the L2-resident integral image -- but none of their addresses is computed per tap.  Layer and octave are template
// parameters, so the corner offsets (vfsms_haar_corner) are constants: a tap is `global_load_dword v, v_lane_offset, s[row base] offset:dx*4`
// -- the row base (y + dy) * pitch is one scalar multiply-add per corner ROW, the lane offset x * 4 is computed once per thread.  (The
// generic k_hessian walks LayerPat tables: its scalar unit issued 3.4 instructions per VALU instruction, PMC round 3.)
// 64 x 4 samples per workgroup; arithmetic and its order are those of k_hessian.
template <int SIZE>
struct HessGatherTap {
    const GAS char *S; uint32_t sw, r0, voff;                      // integral image, its pitch, the sample's row (scalar) and column offset in bytes
    template <int VY, int VX> __device__ __forceinline__ int at() const
    {
        const GAS char *row = S + (size_t)((r0 + (uint32_t)hcorner<SIZE>(VY)) * sw) * 4u;       // scalar: one row base per corner row
        return ((const GAS int32_t *)(row + voff))[hcorner<SIZE>(VX)];
    }
};
template <int OCT, int L>
__device__ __forceinline__ void hessian_coarse_body(const RoiDev &R, const LayerPat *pats, int layers_per_octave, int ti, int tj)
{
    constexpr int STEP = 1 << OCT;
    constexpr int size = (9 + 6 * L) << OCT;
    if (size > R.h || size > R.w) return;
    const int samples_i = 1 + (R.h - size) / STEP, samples_j = 1 + (R.w - size) / STEP;
    const int i = ti * 4 + (int)threadIdx.y, j = tj * 64 + (int)threadIdx.x;
    if (i >= samples_i || j >= samples_j) return;
    const int li = OCT * layers_per_octave + L;
    const LayerPat &P = pats[li];
    const uint32_t sw = (uint32_t)(R.w + 1);
    const GAS char *S = (const GAS char *)R.sum;
    const uint32_t voff = (uint32_t)(j * STEP) * 4u;                 // the lane's column, bytes
    const uint32_t row0 = (uint32_t)(i * STEP);                      // (threadIdx.y is not wave-uniform in general: blockDim = (64, 4) makes it so)
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)row0);
    float w[10];
#pragma unroll
    for (int k = 0; k < 10; k++) w[k] = P.w[k];
    const int margin = (size / 2) / STEP;
    const int lcols = R.w / STEP;
    ((g_f32)R.det[li])[(size_t)(i + margin) * lcols + (j + margin)] = hess_det<size>(HessGatherTap<size>{S, sw, r0, voff}, w);
}

__global__ __launch_bounds__(256) void k_hessian_coarse(const RoiDev *rois, const LayerPat *pats, int layers_per_octave, HessPlan plan, int nrois)
{
    unsigned roi, inner;
    xcd_roi_map(blockIdx.x, (unsigned)(plan.first[plan.noct] * 5), (unsigned)nrois, roi, inner);
    const int l = (int)(inner % 5u);
    int t = (int)(inner / 5u);
    int q = 0;
    while (q + 1 < plan.noct && t >= plan.first[q + 1]) q++;
    t -= plan.first[q];
    const RoiDev &R = rois[roi];
    const int tj = t % plan.tiles_x[q], ti = t / plan.tiles_x[q];
    const int o = plan.o0 + q;
#define HC(O, LL) hessian_coarse_body<O, LL>(R, pats, layers_per_octave, ti, tj)
    if (o == 2) { switch (l) { case 0: HC(2, 0); break; case 1: HC(2, 1); break; case 2: HC(2, 2); break; case 3: HC(2, 3); break; default: HC(2, 4); } }
    else { switch (l) { case 0: HC(3, 0); break; case 1: HC(3, 1); break; case 2: HC(3, 2); break; case 3: HC(3, 3); break; default: HC(3, 4); } }
#undef HC
}

// Octave 2 (step 4), round 6: ROW-STAGED.  The gather kernel above is bound by the texture-address path (TA busy 0.95): the 64 lanes of a
// wave sample columns 16 bytes apart, so every one of its 32 distinct taps per (sample, layer) pulls 1 KB of cache lines through the
// L1 for 256 useful bytes, and a (tile + wavelet) window of the octave does not fit LDS (120-200 KB).  But a layer's taps touch only
// TEN integral rows per sample row (4 i + corner(v), v = 0..9: Dx uses rows 2, 7, Dy 0, 3, 6, 9, Dxy 1, 4, 5, 8 of the scaled
// pattern), and every column of those rows inside the tile's span is used by some sample.  So one workgroup = (layer, 2 sample
// rows, 128 samples): it stages the 20 row segments [4 j0, 4 (j0 + 127) + size] with coalesced 16-byte loads (51 KB for
// the 132-px layer, three workgroups per CU), each row as four column planes (x & 3), and every tap is one conflict-free
// ds_read_b32 at an immediate offset: lane jl needs column 4 jl + dx = element jl + (dx >> 2) of plane dx & 3 -- unit stride
// across the lanes (a linear row would be read with a stride of four dwords: four lanes per bank).  Loaded bytes per sample drop from 32 x 16 (scattered) to ~40 x 4 (coalesced).  Arithmetic and its order
// are those of k_hessian.
template <int SIZE, int W4>
struct HessRowsTap {
    const int32_t *base;                                           // the lane's element of row slot 0 / plane 0
    template <int VY, int VX> __device__ __forceinline__ int at() const
    {
        constexpr int dx = hcorner<SIZE>(VX);
        return base[VY * (4 * W4) + (dx & 3) * W4 + (dx >> 2)];   // row slot = pattern row, plane = column phase
    }
};
template <int L>
__device__ __forceinline__ void hessian_rows2_body(const RoiDev &R, const LayerPat *pats, int layers_per_octave, int ti, int tj, int32_t *lds)
{
    constexpr int STEP = 4;
    constexpr int size = (9 + 6 * L) << 2;
    constexpr int W4 = 128 + size / 4;                              // 16-byte chunks per staged row (size is a multiple of 4)
    constexpr int RP = 4 * W4;                                      // ints per row slot: four column planes of W4
    if (size > R.h || size > R.w) return;
    const int samples_i = 1 + (R.h - size) / STEP, samples_j = 1 + (R.w - size) / STEP;
    const int i0 = ti * 2, j0 = tj * 128;
    if (i0 >= samples_i || j0 >= samples_j) return;                 // (whole workgroup)
    const int sw = R.w + 1, sh = R.h + 1;
    const int lane = threadIdx.x, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
    const GAS char *S = (const GAS char *)R.sum;
    const int x0 = j0 * STEP;
    {
        constexpr int NC = (W4 + 63) / 64;
        uint32_t xo[NC]; bool ok[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int cc = lane + 64 * c, x = x0 + 4 * cc;
            ok[c] = cc < W4 && x <= sw - 1;                         // a chunk that starts beyond the row holds nothing a valid sample reads
            xo[c] = (uint32_t)min(x, sw - 1) * 4u;
        }
        // 20 row slots, 5 per wave: slot = si * 10 + v
        int4u t[5][NC];
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const int slot = wave + 4 * r;
            const int si = slot >= 10 ? 1 : 0, v = slot - 10 * si;
            const int y = min(STEP * (i0 + si) + (2 * size * v + 9) / 18, sh - 1);
            const GAS char *row = S + (size_t)y * sw * 4u;
#pragma unroll
            for (int c = 0; c < NC; c++) t[r][c] = *(const GAS int4u *)(row + xo[c]);
        }
#pragma unroll
        for (int r = 0; r < 5; r++) {
            int32_t *dst = lds + (wave + 4 * r) * RP + lane;
#pragma unroll
            for (int c = 0; c < NC; c++)
                if (ok[c]) { dst[64 * c] = t[r][c].x; dst[W4 + 64 * c] = t[r][c].y; dst[2 * W4 + 64 * c] = t[r][c].z; dst[3 * W4 + 64 * c] = t[r][c].w; }
        }
    }
    __syncthreads();
    const int si = wave >> 1, jl = (wave & 1) * 64 + lane;
    const int i = i0 + si, j = j0 + jl;
    if (i >= samples_i || j >= samples_j) return;
    const int li = 2 * layers_per_octave + L;
    const LayerPat &P = pats[li];
    const int32_t *base = lds + si * 10 * RP + jl;
    float w[10];
#pragma unroll
    for (int k = 0; k < 10; k++) w[k] = P.w[k];
    const int margin = (size / 2) / STEP;
    const int lcols = R.w / STEP;
    ((g_f32)R.det[li])[(size_t)(i + margin) * lcols + (j + margin)] = hess_det<size>(HessRowsTap<size, W4>{base}, w);
}

#define HESS_ROWS2_LDS_INTS (20 * 4 * (128 + 33))
__global__ __launch_bounds__(256) void k_hessian_rows2(const RoiDev *rois, const LayerPat *pats, int layers_per_octave, int tiles_i, int tiles_j, int nrois)
{
    __shared__ int32_t lds[HESS_ROWS2_LDS_INTS];
    unsigned roi, inner;
    xcd_roi_map(blockIdx.x, (unsigned)(tiles_i * tiles_j * 5), (unsigned)nrois, roi, inner);
    const int l = (int)(inner % 5u);
    const int t = (int)(inner / 5u);
    const int tj = t % tiles_j, ti = t / tiles_j;
    const RoiDev &R = rois[roi];
    switch (l) {
    case 0: hessian_rows2_body<0>(R, pats, layers_per_octave, ti, tj, lds); break;
    case 1: hessian_rows2_body<1>(R, pats, layers_per_octave, ti, tj, lds); break;
    case 2: hessian_rows2_body<2>(R, pats, layers_per_octave, ti, tj, lds); break;
    case 3: hessian_rows2_body<3>(R, pats, layers_per_octave, ti, tj, lds); break;
    default: hessian_rows2_body<4>(R, pats, layers_per_octave, ti, tj, lds);
    }
}

// ---------------------------------------------------------------------------------------------------
// K3 3x3x3 non-maximum suppression + interpolateKeypoint (Cramer's rule in float, as
//   Matx33f::solve(DECOMP_LU) does) + atomic append
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool interpolate_keypoint(const float N9[3][9], int dx, int dy, int ds, Cand &kpt)
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
    float d = det;
    if (d != 0) {
        d = 1 / d;
        x0 = d * (b0 * (a11 * a22 - a12 * a21) - a01 * (b1 * a22 - a12 * b2) + a02 * (b1 * a21 - a11 * b2));
        x1 = d * (a00 * (b1 * a22 - a12 * b2) - b0 * (a10 * a22 - a12 * a20) + a02 * (a10 * b2 - b1 * a20));
        x2 = d * (a00 * (a11 * b2 - b1 * a21) - a01 * (a10 * b2 - b1 * a20) + b0 * (a10 * a21 - a11 * a20));
    }
    bool ok = (x0 != 0 || x1 != 0 || x2 != 0) && fabsf(x0) <= 1 && fabsf(x1) <= 1 && fabsf(x2) <= 1;
    if (ok) {
        kpt.x += x0 * dx;
        kpt.y += x1 * dy;
        kpt.size = (float)cv_round_f(kpt.size + x2 * ds);
    }
    return ok;
}

// One launch covers every octave: blockIdx.x walks the 64 x 64-cell tiles of octave 0, then those of octave 1, ... (NmsPlan), so
// the small coarse octaves fill the tail of the fine one instead of paying a launch each; blockIdx.y = roi * n_middle + middle layer.
// A wave streams NMS_RW rows of a 64-column strip with the three rows it needs in registers (the +-1 column taps are unaligned
// loads of the same lines, L1 hits): no LDS tile, no barrier in the scan.  A cell above the threshold that beats its 8 own-layer
// neighbours -- a fraction of a per cent of the cells -- is queued in the wave's LDS queue (ballot + prefix count); afterwards the
// queue is examined one cell per lane against the layers below and above, so the 18 extra taps and the 3 x 3 solve run on full waves.
#define NMS_RW 16
#ifndef NMS_PF
#define NMS_PF 3               // rows in flight ahead of the compared row
#endif
#define NMS_TH (4 * NMS_RW)
#define NMS_LOCAL 192
struct NmsPlan { int noct; int first[VFSMS_MAX_OCTAVES + 1]; int tiles_x[VFSMS_MAX_OCTAVES]; };

__global__ __launch_bounds__(256) void k_nms(const RoiDev *rois, const LayerPat *pats, int layers_per_octave,
                                             int n_middle, NmsPlan plan, float hessianThreshold)
{
    int octave = 0;
    while (octave + 1 < plan.noct && (int)blockIdx.x >= plan.first[octave + 1]) octave++;
    const int tix = (int)blockIdx.x - plan.first[octave];
    const int bx = tix % plan.tiles_x[octave], by = tix / plan.tiles_x[octave];
    const int roi = blockIdx.y / n_middle;
    const int l = 1 + blockIdx.y % n_middle;
    const int li = octave * layers_per_octave + l;
    const RoiDev &R = rois[roi];
    const LayerPat &P = pats[li];
    const int ss = P.step, size = P.size;
    const int lrows = R.h / ss, lcols = R.w / ss;
    const int margin = (pats[li + 1].size / 2) / ss + 1;
    if (pats[li + 1].size > R.h || pats[li + 1].size > R.w) return;   // upper layer not computed: nothing readable
    const int lane = threadIdx.x, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.y), tid = wave * 64 + lane;
    const int j0 = margin + bx * 64, i0 = margin + by * NMS_TH;
    if (j0 >= lcols - margin || i0 >= lrows - margin) return;        // (whole workgroup)
    __shared__ unsigned short queue[4][NMS_RW * 64];
    __shared__ int cn, cbase;
    __shared__ Cand local[NMS_LOCAL];
    if (tid == 0) cn = 0;
    __syncthreads();
    g_cf32 d2 = (g_cf32)R.det[li];
    const int st = lcols;
    const int j = j0 + lane;
    const int ia = i0 + wave * NMS_RW;
    int nq = 0;
    if (ia < lrows - margin) {
        // evaluated cells have all 8 neighbours inside the layer; the clamps only keep the loads of the other lanes in bounds
        const int jl = min(j - 1, lcols - 1), jc = min(j, lcols - 1), jr = min(j + 1, lcols - 1);
        const bool jev = j < lcols - margin;
        // Round 6: the rows are requested NMS_PF rows AHEAD of the row they are compared in.  The scan was one dependent round trip per
        // row (three loads, compare, next row): a workgroup needs 16 of them and a CU holds eight workgroups, which is what the launch
        // lasted (PMC: the waves waited 82 % of their cycles).  rw[k] = row ia - 1 + k of the strip (18 rows: the 16 scanned + one above, one below).
        float rw[NMS_RW + 2][3];
        auto load_row = [&](int k) {
            g_cf32 rp = d2 + (size_t)min(ia - 1 + k, lrows - 1) * st;
            rw[k][0] = rp[jl]; rw[k][1] = rp[jc]; rw[k][2] = rp[jr];
        };
#pragma unroll
        for (int k = 0; k < 2 + NMS_PF; k++) load_row(k);
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r < NMS_RW; r++) {
            const int i = ia + r;
            if (r + 2 + NMS_PF < NMS_RW + 2) load_row(r + 2 + NMS_PF);
            const float *up = rw[r], *mid = rw[r + 1], *dn = rw[r + 2];
            const float v = mid[1];
            const bool c2 = jev && i < lrows - margin && v > hessianThreshold &&
                            v > up[0] && v > up[1] && v > up[2] && v > mid[0] && v > mid[2] && v > dn[0] && v > dn[1] && v > dn[2];
            const unsigned long long m = __ballot(c2);
            if (m) {
                if (c2) queue[wave][nq + __popcll(m & below)] = (unsigned short)((r << 8) | lane);
                nq += __popcll(m);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < nq; e += 64) {
        const int qe = queue[wave][e];
        const int i = ia + (qe >> 8), jj = j0 + (qe & 255);
        g_cf32 d1 = (g_cf32)R.det[li - 1] + (size_t)i * lcols + jj;
        g_cf32 dm = d2 + (size_t)i * lcols + jj;
        g_cf32 d3 = (g_cf32)R.det[li + 1] + (size_t)i * lcols + jj;
        const float val0 = dm[0];
        float N9[3][9] = {
            { d1[-st - 1], d1[-st], d1[-st + 1], d1[-1], d1[0], d1[1], d1[st - 1], d1[st], d1[st + 1] },
            { dm[-st - 1], dm[-st], dm[-st + 1], dm[-1], val0, dm[1], dm[st - 1], dm[st], dm[st + 1] },
            { d3[-st - 1], d3[-st], d3[-st + 1], d3[-1], d3[0], d3[1], d3[st - 1], d3[st], d3[st + 1] } };
        bool is_max = true;
#pragma unroll
        for (int b = 0; b < 9; b++) is_max = is_max && (val0 > N9[0][b]) && (val0 > N9[2][b]);
        if (!is_max) continue;
        const int sum_i = ss * (i - (size / 2) / ss);
        const int sum_j = ss * (jj - (size / 2) / ss);
        Cand cd;
        cd.y = sum_i + (size - 1) * 0.5f;
        cd.x = sum_j + (size - 1) * 0.5f;
        cd.size = (float)size;
        cd.response = val0;
        cd.octave = octave;
        cd.class_id = 0;                                   // sign of the trace: filled in by the ordering kernels (cand_class_id)
        cd.layer = li; cd.i = i; cd.j = jj;
        const int ds = size - pats[li - 1].size;
        if (!interpolate_keypoint(N9, ss, ss, ds, cd)) continue;
        const int slot = atomicAdd(&cn, 1);
        if (slot < NMS_LOCAL) local[slot] = cd;
        else {                                           // local list full (never seen on real images): direct append
            const int pos = atomicAdd(&R.counters[0], 1);
            if (pos < R.cap) R.cand[pos] = cd; else R.counters[2] = 1;
        }
    }
    // one global reservation per workgroup instead of one same-address atomic per candidate
    __syncthreads();
    const int nloc = min(cn, NMS_LOCAL);
    if (nloc == 0) return;
    if (tid == 0) cbase = atomicAdd(&R.counters[0], nloc);
    __syncthreads();
    const int base = cbase;
    if (base + nloc > R.cap) { if (tid == 0) R.counters[2] = 1; }   // overflow: reported as VFSMS_ERR_CAPACITY by the host
    constexpr int CW = sizeof(Cand) / 4;
    const int nfit = max(0, min(nloc, R.cap - base));
    uint32_t *dst = (uint32_t *)(R.cand + base);
    const uint32_t *srcw = (const uint32_t *)local;
    for (int w = tid; w < nfit * CW; w += 256) dst[w] = srcw[w];
}

// ---------------------------------------------------------------------------------------------------
// keypoint ordering: std::sort(KeypointGreater) == rank by counting (N^2 compares through LDS tiles).
//   order: response desc, size desc, octave desc, y DESC, x asc (upstream surf.cpp KeypointGreater), then (layer, i, j)
//   asc to make it total.
// ---------------------------------------------------------------------------------------------------
struct SortKey { unsigned long long k1, k2, k3; };
// k1 (descending): response bits (positive floats order like their bit patterns) then integer size then octave;
// k2 (ascending): the order-REVERSING image of y (y descending) then the order-preserving image of x;
// k3 (ascending): (layer, i, j), unique per candidate.

__device__ __forceinline__ uint32_t ord_f32(float f)
{
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ SortKey make_key(const Cand &c)
{
    SortKey k;
    k.k1 = ((unsigned long long)__float_as_uint(c.response) << 32) | (uint32_t)((((int)c.size) << 4) | c.octave);
    k.k2 = ((unsigned long long)(uint32_t)~ord_f32(c.y) << 32) | ord_f32(c.x);
    k.k3 = ((unsigned long long)(uint32_t)c.layer << 32) | (((uint32_t)c.i << 16) | (uint32_t)c.j);
    return k;
}

__global__ __launch_bounds__(256) void k_rank_sort(const RoiDev *rois, const LayerPat *pats)
{
    // one workgroup ranks 64 candidates; its 4 waves each scan a quarter of every 256-key LDS tile
    const RoiDev &R = rois[blockIdx.y];
    const int n = min(R.counters[0], R.cap);
    if ((int)(blockIdx.x * 64) >= n) return;
    const int lane = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int idx = blockIdx.x * 64 + lane;
    __shared__ unsigned long long t1[256], t2[256], t3[256];
    __shared__ int partial[4][64];
    Cand me;
    SortKey mk; mk.k1 = mk.k2 = mk.k3 = 0;
    if (idx < n) { me = R.cand[idx]; mk = make_key(me); }
    int rank = 0;
    for (int base = 0; base < n; base += 256) {
        int t = base + threadIdx.x;
        if (t < n) { SortKey k = make_key(R.cand[t]); t1[threadIdx.x] = k.k1; t2[threadIdx.x] = k.k2; t3[threadIdx.x] = k.k3; }
        __syncthreads();
        const int k0 = part * 64, k1e = min(k0 + 64, n - base);
        if (idx < n) {
            bool tie = false;
#pragma unroll 8
            for (int k = k0; k < k1e; k++) {
                const unsigned long long a = t1[k];          // LDS broadcast read
                rank += (a > mk.k1) ? 1 : 0;
                tie |= (a == mk.k1);
            }
            if (tie)                                           // rare: self, or equal response/size/octave
                for (int k = k0; k < k1e; k++)
                    if (t1[k] == mk.k1) rank += (t2[k] < mk.k2 || (t2[k] == mk.k2 && t3[k] < mk.k3)) ? 1 : 0;
        }
        __syncthreads();
    }
    partial[part][lane] = rank;
    __syncthreads();
    if (part == 0 && idx < n) {
        rank = partial[0][lane] + partial[1][lane] + partial[2][lane] + partial[3][lane];
        vfsms_keypoint kp;
        kp.x = me.x; kp.y = me.y; kp.size = me.size; kp.angle = -1.f; kp.response = me.response;
        kp.octave = me.octave; kp.class_id = cand_class_id(R, pats, me);
        R.kps[rank] = kp;
    }
}

// The same order by BUCKETS instead of all-pairs counting (k_rank_sort above: n^2 key compares, 190 us per 16-ROI launch): one 1024-thread
// workgroup per ROI histograms the candidates over 1024 buckets of the response's float bits (exponent + 5 mantissa bits; responses
// are > hessianThreshold), scans the histogram from the top (descending order), scatters (key, index) records bucket by bucket into
// scratch (the patch buffer, unused until the descriptor stage) and ranks every candidate against its own bucket only -- a few dozen
// full KeypointGreater compares instead of n (k_bucket_rank, on the whole chip).  The result is the identical permutation: rank = (candidates in higher buckets) +
// (candidates of the same bucket that sort before it), and bucket order is response order.
#define SORT_BUCKETS 1024
struct SortRec { unsigned long long k1, k2, k3; int idx, pad; };
__device__ __forceinline__ int sort_bucket(unsigned long long k1)
{
    const int b = (int)((uint32_t)(k1 >> 32) >> 18) - (133 << 5);     // float bits >> 18: exponent and 5 mantissa bits; 2^6 <= response
    return min(max(b, 0), SORT_BUCKETS - 1);
}
__global__ __launch_bounds__(1024) void k_bucket_sort(const RoiDev *rois)
{
    const RoiDev &R = rois[blockIdx.x];
    const int n = min(R.counters[0], R.cap);
    if (n <= 0) return;
    __shared__ int hist[SORT_BUCKETS], base[SORT_BUCKETS], cursor[SORT_BUCKETS], wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    hist[tid] = 0; cursor[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[sort_bucket(make_key(R.cand[i]).k1)], 1);
    __syncthreads();
    {   // base[b] = candidates in buckets above b: inclusive scan over the reversed histogram minus the own count
        const int rb = SORT_BUCKETS - 1 - tid;
        const int v = hist[rb];
        int incl = wave_incl_scan(v);
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int off = 0;
        for (int k = 0; k < wid; k++) off += wsum[k];
        base[rb] = off + incl - v;
    }
    __syncthreads();
    SortRec *S = reinterpret_cast<SortRec *>(R.patch);                // cap * 464 B >= cap * 32 B
    for (int i = tid; i < n; i += 1024) {
        const SortKey k = make_key(R.cand[i]);
        const int b = sort_bucket(k.k1);
        const int pos = base[b] + atomicAdd(&cursor[b], 1);
        SortRec r; r.k1 = k.k1; r.k2 = k.k2; r.k3 = k.k3; r.idx = i; r.pad = 0;
        S[pos] = r;
    }
    // bucket bounds for the ranking kernel (keep_pos / order are free until the descriptor stage; the launcher checks cap >= SORT_BUCKETS)
    R.keep_pos[tid] = base[tid]; R.order[tid] = hist[tid];
}

// rank of every candidate inside its bucket (a few dozen to a few hundred full KeypointGreater compares), 256 candidates per workgroup
__global__ __launch_bounds__(256) void k_bucket_rank(const RoiDev *rois, const LayerPat *pats)
{
    const RoiDev &R = rois[blockIdx.y];
    const int n = min(R.counters[0], R.cap);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const SortRec *S = reinterpret_cast<const SortRec *>(R.patch);
    const SortRec me = S[p];
    const int b = sort_bucket(me.k1);
    const int lo = R.keep_pos[b], hi = lo + R.order[b];
    int rank = lo;
    for (int q = lo; q < hi; q++) {
        const unsigned long long a1 = S[q].k1;
        if (a1 > me.k1) rank++;
        else if (a1 == me.k1) {
            const unsigned long long a2 = S[q].k2, a3 = S[q].k3;
            rank += (a2 < me.k2 || (a2 == me.k2 && a3 < me.k3)) ? 1 : 0;
        }
    }
    const Cand c = R.cand[me.idx];
    vfsms_keypoint kp;
    kp.x = c.x; kp.y = c.y; kp.size = c.size; kp.angle = -1.f; kp.response = c.response;
    kp.octave = c.octave; kp.class_id = cand_class_id(R, pats, c);
    R.kps[rank] = kp;
}

// ---------------------------------------------------------------------------------------------------
// K4 orientation (SURFInvoker, upright == 0): 128 threads per keypoint
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x)   // core atan_f32 polynomial, degrees
{
    const float s = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
    const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
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

// K4 dominant orientation, ORI_KP keypoints per workgroup of 128 lanes each (eight per 1024 threads until round 6, now two).  The three phases have different widths -- 113 gradient
// samples, 72 sliding windows, one argmax per keypoint -- and a workgroup per keypoint left most lanes of its second wave idle in the
// window phase (72 windows = one wave + 8 lanes) and one lane walking the 72 moduli.  Here phase 1 gives every keypoint 128 lanes
// (sample = lane), phase 2 packs the 8 x 72 windows into nine full waves, phase 3 reduces each keypoint's 72 moduli in one wave
// (first maximum in window order, as the reference's strict `>` scan keeps it).  The sums visit the samples in index order.
//
// Round 6.  Phase 1: the two gradient wavelets of resizeHaarPattern (dx_s = {{0,0,2,4,-1},{2,0,4,4,1}}, dy_s = {{0,0,4,2,1},{0,2,4,4,-1}})
// share a 3 x 3 corner lattice (x, y in {0, r2, r4}): eight distinct integral taps, and their four weights are +-1 / (r2 * r4) and
// +-1 / ((r4 - r2) * r4) -- the float product is commutative and IEEE division is sign-symmetric, so TWO divisions per sample give the
// reference's four quotients bit for bit.  Phase 2: a sample's 72-bit window membership is kept as NINE BYTES (byte g = windows 8 g ..
// 8 g + 7), sample-minor, so the lane of window w reads the bytes of four consecutive samples as one dword, isolates its bit in all four
// with one shift and one mask, and each byte becomes the float 0 / 1 by v_cvt_f32_ubyteN; the masked accumulation of (x, y) is one packed
// fma per sample -- fma(v, 1, s) = s + v and fma(v, 0, s) = s exactly (s is never -0: it starts at +0 and x + (-x) rounds to +0) -- 2.5
// instructions per (sample, window) instead of 6.
#ifndef ORI_KP
#define ORI_KP 2                // keypoints per workgroup (128 lanes each): 8 -> 0.571, 4 -> 0.539, 2 -> 0.534, 1 -> 0.547 ms on the 16-pair batch
#endif
typedef float ori_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void orientation_block(const RoiDev &R, const SurfTables *T, const int k0, const int n, int upright)
{
    __shared__ __attribute__((aligned(16))) float2 XY[ORI_KP][128];      // weighted gradient (x, y) of every sample, interleaved: phase 2 reads pairs
    __shared__ __attribute__((aligned(16))) uint8_t Zb[ORI_KP][9][128];   // byte g of the window membership of every sample (SurfTables::oriMask rows)
    __shared__ float mod_s[ORI_KP][72], sx_s[ORI_KP][72], sy_s[ORI_KP][72];
    __shared__ int nvalid[ORI_KP], state[ORI_KP];              // state: 0 compute, 1 done (deleted / upright / beyond n)
    const int tid = threadIdx.x;
    const int kq = __builtin_amdgcn_readfirstlane(tid >> 7), t = tid & 127;      // keypoint slot: wave-uniform, kept scalar
    if (tid < ORI_KP) { nvalid[tid] = 0; state[tid] = 0; }
    __syncthreads();
    {
        const int k = k0 + kq;
        int valid = 0;
        uint32_t m0 = 0, m1 = 0, m2 = 0;
        float2 xy = make_float2(0.f, 0.f);
        if (k < n) {
            const vfsms_keypoint kp = R.kps[k];
            const float s = kp.size * 1.2f / 9.0f;
            const int gws = 2 * cv_round_f(2 * s);
            const int srows = R.h + 1, scols = R.w + 1, sw = R.w + 1;
            if (srows < gws || scols < gws) {                  // gradient wavelet larger than the image: delete
                if (t == 0) { R.kps[k].size = -1.f; state[kq] = 1; }
            } else if (upright) {
                if (t == 0) { R.kps[k].angle = 270.f; state[kq] = 1; }
            } else if (t < T->nOriSamples) {
                int x = cv_round_f(kp.x + T->aptx[t] * s - (float)(gws - 1) / 2);
                int y = cv_round_f(kp.y + T->apty[t] * s - (float)(gws - 1) / 2);
                if (!(y < 0 || y >= srows - gws || x < 0 || x >= scols - gws)) {
                    // resizeHaarPattern for the 4-unit gradient wavelets + calcHaarPattern (2 boxes each)
                    const float ratio = (float)gws / 4;
                    const int r2 = cv_round_f(ratio * 2), r4 = cv_round_f(ratio * 4);
                    g_ci32 p0 = (g_ci32)R.sum + (size_t)y * sw + x;
                    g_ci32 p2 = p0 + (size_t)r2 * sw, p4 = p0 + (size_t)r4 * sw;
                    const int a00 = p0[0], a02 = p0[r2], a04 = p0[r4];          // a<row><column> of the corner lattice
                    const int a20 = p2[0], a24 = p2[r4];
                    const int a40 = p4[0], a42 = p4[r2], a44 = p4[r4];
                    // gws is even, so r2 = gws / 2 and r4 = gws exactly and r4 - r2 = r2: the two boxes of a wavelet have ONE area -- one
                    // division.  (A host-built table of the reciprocals measured 1.4 % SLOWER than the division: profiles/r06_ab_detect_stage.txt.)
                    const float wA = 1.f / ((float)r2 * (float)r4);
                    float wB = wA;
                    if (r4 - r2 != r2) wB = 1.f / ((float)(r4 - r2) * (float)r4);
                    const int vx0 = a00 + a42 - a40 - a02, vx1 = a02 + a44 - a42 - a04;     // dx: {0,0,r2,r4} weight -wA, {r2,0,r4,r4} weight +wB
                    const int vy0 = a00 + a24 - a20 - a04, vy1 = a20 + a44 - a40 - a24;     // dy: {0,0,r4,r2} weight +wA, {0,r2,r4,r4} weight -wB
                    double dxd = 0; dxd += (double)(-((float)vx0 * wA)); dxd += (double)((float)vx1 * wB);
                    double dyd = 0; dyd += (double)((float)vy0 * wA); dyd += (double)(-((float)vy1 * wB));
                    const float vx = (float)dxd, vy = (float)dyd;
                    const float aw = T->aptw[t];
                    const float xx = vx * aw, yy = vy * aw;
                    xy = make_float2(xx, yy);
                    const int ang = cv_round_f(fast_atan2_deg(yy, xx));     // cv::phase(X, Y, angle, true) then cvRound
                    const uint32_t *mrow = T->oriMask[min(max(ang, 0), 360)];
                    m0 = mrow[0]; m1 = mrow[1]; m2 = mrow[2];
                    valid = 1;
                }
            }
        } else if (t == 0) state[kq] = 1;
        // a sample outside the image (or past nOriSamples) belongs to no window and carries a zero gradient
        XY[kq][t] = xy;
        Zb[kq][0][t] = (uint8_t)m0; Zb[kq][1][t] = (uint8_t)(m0 >> 8); Zb[kq][2][t] = (uint8_t)(m0 >> 16); Zb[kq][3][t] = (uint8_t)(m0 >> 24);
        Zb[kq][4][t] = (uint8_t)m1; Zb[kq][5][t] = (uint8_t)(m1 >> 8); Zb[kq][6][t] = (uint8_t)(m1 >> 16); Zb[kq][7][t] = (uint8_t)(m1 >> 24);
        Zb[kq][8][t] = (uint8_t)m2;
        const unsigned long long m = __ballot(valid);
        if ((tid & 63) == 0 && m) atomicAdd(&nvalid[kq], __popcll(m));
    }
    __syncthreads();
    if (tid < ORI_KP * 72) {
        const int q = tid / 72, w = tid - q * 72;
        if (state[q] == 0 && nvalid[q] > 0) {
            const int nori = T->nOriSamples;                  // <= 113: the groups of eight below cover 120 samples, the slots past nori are zero
            const uint8_t *zrow = Zb[q][w >> 3];
            const int bit = w & 7;
            ori_f2 acc = {0.f, 0.f};
            for (int j = 0; j < nori; j += 8) {
                const uint2 z8 = *reinterpret_cast<const uint2 *>(&zrow[j]);
                const uint32_t za = (z8.x >> bit) & 0x01010101u, zb = (z8.y >> bit) & 0x01010101u;
                const float4 p01 = *reinterpret_cast<const float4 *>(&XY[q][j]), p23 = *reinterpret_cast<const float4 *>(&XY[q][j + 2]);
                const float4 p45 = *reinterpret_cast<const float4 *>(&XY[q][j + 4]), p67 = *reinterpret_cast<const float4 *>(&XY[q][j + 6]);
#define ORI_ACC(Z, N, P, XC, YC) do { float mf; asm("v_cvt_f32_ubyte" #N " %0, %1" : "=v"(mf) : "v"(Z)); \
                                           acc = __builtin_elementwise_fma((ori_f2){P.XC, P.YC}, (ori_f2){mf, mf}, acc); } while (0)
                ORI_ACC(za, 0, p01, x, y); ORI_ACC(za, 1, p01, z, w); ORI_ACC(za, 2, p23, x, y); ORI_ACC(za, 3, p23, z, w);
                ORI_ACC(zb, 0, p45, x, y); ORI_ACC(zb, 1, p45, z, w); ORI_ACC(zb, 2, p67, x, y); ORI_ACC(zb, 3, p67, z, w);
#undef ORI_ACC
            }
            const float sumx = acc.x, sumy = acc.y;
            mod_s[q][w] = sumx * sumx + sumy * sumy;
            sx_s[q][w] = sumx; sy_s[q][w] = sumy;
        }
    }
    __syncthreads();
    {
        const int q = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        if (q < ORI_KP && state[q] == 0) {
            const int k = k0 + q;
            if (nvalid[q] == 0) {
                if (lane == 0) R.kps[k].size = -1.f;
            } else {
                // the reference keeps the FIRST window whose modulus exceeds every earlier one, starting from 0: a modulus that is
                // not > 0 (zero or NaN) never wins; ties go to the lower window index
                float bm = mod_s[q][lane]; int bi = lane;
                if (!(bm > 0.f)) bm = 0.f;
                if (lane < 8) { float m2 = mod_s[q][64 + lane]; if (m2 > bm) { bm = m2; bi = 64 + lane; } }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const float om = __shfl_xor(bm, off, 64); const int oi = __shfl_xor(bi, off, 64);
                    if (om > bm || (om == bm && oi < bi)) { bm = om; bi = oi; }
                }
                if (lane == 0) {
                    float bestx = 0, besty = 0;
                    if (bm > 0.f) { bestx = sx_s[q][bi]; besty = sy_s[q][bi]; }
                    R.kps[k].angle = fast_atan2_deg(-besty, bestx);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// K5 descriptor: one 256-thread workgroup per keypoint.
//   The rotated win_size^2 window is never materialised: each of the 21x21 INTER_AREA output pixels
//   re-derives the bilinear samples of its own source cell in exactly the accumulation order of
//   cv::resize's three area paths.  Row origins (start_x/start_y) are running float sums in the
//   reference, so lane 0 produces them sequentially into LDS first.
// ---------------------------------------------------------------------------------------------------
#ifdef VFSMS_DESC_TIMING
__device__ unsigned long long g_desc_cycles[8];
#define DT_MARK(ph) do { if (threadIdx.x == 0) { unsigned long long _n = clock64(); atomicAdd(&g_desc_cycles[ph], _n - _t0); _t0 = _n; } } while (0)
#define DT_START unsigned long long _t0 = clock64()
extern "C" int vfsms_debug_desc_cycles(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_desc_cycles), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -3; }
__device__ unsigned long long g_desc_trips[4];      // [0] interior-strip trips, [1] border-strip trips that are all inside, [2] border trips (per-sample path), [3] keypoints
#define DT_TRIP(n) do {} while (0)
__device__ unsigned long long g_desc_unit_cycles[4];  // wave-cycles inside [0] interior units, [1] border units, [2] stage_rows as a whole (per wave)
#define DT_UNIT_BEGIN unsigned long long _u0 = clock64()
#define DT_UNIT_END(n) do { _uacc[n] += clock64() - _u0; _ucnt[n]++; } while (0)
extern "C" int vfsms_debug_desc_unit_cycles(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_desc_unit_cycles), sizeof(unsigned long long) * 4) == hipSuccess ? 0 : -3; }
extern "C" int vfsms_debug_desc_trips(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_desc_trips), sizeof(unsigned long long) * 4) == hipSuccess ? 0 : -3; }
#else
#define DT_MARK(ph) do {} while (0)
#define DT_START do {} while (0)
#define DT_TRIP(n) do {} while (0)
#define DT_UNIT_BEGIN do {} while (0)
#define DT_UNIT_END(n) do {} while (0)
#endif
#ifndef VFSMS_EXP
#define VFSMS_EXP 0
#endif
#ifndef DESC_WBUF
#define DESC_WBUF 15360       // (round 6: 16384 -> 15360 makes room for the sixth workgroup of a CU: 6 x 26.2 KB of LDS)
#endif                            // LDS bytes for the staged descriptor window (win <= 128) / band chunk
// start, start + d, (start + d) + d, ... : the reference's running float sum (one rounding per step, so the chain cannot be split), four
// links per trip -- the trip overhead (counter, compare, branch, address) was four fifths of the instructions of this single-lane loop
__device__ __forceinline__ void origin_chain(float *row, float s, const float d, const int n)
{
    int i = 0;
    for (; i + 4 <= n; i += 4) {
        const float a = s, b = a + d, c = b + d, e = c + d;
        s = e + d;
        row[i] = a; row[i + 1] = b; row[i + 2] = c; row[i + 3] = e;
    }
    for (; i < n; i++, s += d) row[i] = s;
}

struct WinGeom {
    int win; float sin_dir, cos_dir;
    int h, w, stride; g_cu8 img;
    g_cu8 pair;                       // RoiDev::pair as bytes: the 2 x 2 taps of a sample are ONE dword at 2 * (y * w + x)
    int upright, usx, usy;            // upright: integer lattice origin (start_x, start_y)
};

// One bilinear sample of the rotated window, WIN[i][j] of SURFInvoker.  px/py are the reference's double
// pixel_x / pixel_y (start + j * step, exact in double).  Fast path for the interior: floor by truncation
// (coordinates are non-negative there) and the fractional parts straight from v_fract_f64 (x - floor(x) is exact).
__device__ __forceinline__ int win_sample_xy(const WinGeom &G, double px, double py)
{
    const int ncols1 = G.w - 1, nrows1 = G.h - 1;
    const int ix = (int)px, iy = (int)py;                        // trunc; == floor when px, py >= 0
    if (px >= 0.0 && py >= 0.0 && ix < ncols1 && iy < nrows1) {
        const float a = (float)__builtin_amdgcn_fract(px), b = (float)__builtin_amdgcn_fract(py);
        g_cu8 p = G.img + (size_t)iy * G.stride + ix;
        const float v = p[0] * (1.f - a) * (1.f - b) + p[1] * a * (1.f - b) + p[G.stride] * (1.f - a) * b + p[G.stride + 1] * a * b;
        return (int)(uint8_t)cv_round_f(v);
    }
    const int x = min(max(cv_round_d(px), 0), ncols1);
    const int y = min(max(cv_round_d(py), 0), nrows1);
    return (int)G.img[(size_t)y * G.stride + x];
}
__device__ __forceinline__ int win_sample_upright(const WinGeom &G, int i, int j)
{
    // WIN[i][j] = img[clamp(start_y - j)][clamp(start_x + i)]
    const int x = min(max(G.usx + i, 0), G.w - 1);
    const int y = min(max(G.usy - j, 0), G.h - 1);
    return (int)G.img[(size_t)y * G.stride + x];
}

// Stage rows [r0, r0 + nrows) x all `win` columns of the window into LDS (row-major, pitch win).  Each wave sweeps
// 8-row strips left to right in 8x8-lane tiles (a wave's byte gathers touch ~10 cache lines instead of ~45), FOUR
// tiles per trip: the 16 byte loads of a lane are issued back to back behind ONE wave-uniform interior test, so the
// L2 latency is paid once per four samples.  (Measured: the kernel as a whole is bound by instruction issue at 5 workgroups/CU;
// changes of tile shape, ILP depth or occupancy beyond that left its time unchanged -- DESIGN.md section 9.)
typedef float float2v __attribute__((ext_vector_type(2)));
// one bilinear sample from the dword of the row-pair image (bytes t00, t10, t01, t11), in the reference's operation order
// t00 (1-a) (1-b) + t01 a (1-b) + t10 (1-a) b + t11 a b; the two products that share a factor go through one v_pk_mul_f32
__device__ __forceinline__ float bilinear_pk(uint32_t top, float a, float b)
{
    const float na = 1.f - a, nb = 1.f - b;
    const float2v w = {na, a};
    const float2v r0 = {(float)(top & 0xff), (float)((top >> 16) & 0xff)};
    const float2v r1 = {(float)((top >> 8) & 0xff), (float)(top >> 24)};
    float2v p = r0 * w, q = r1 * w;
    const float2v nb2 = {nb, nb}, b2 = {b, b};
    p = p * nb2; q = q * b2;
    return ((p.x + p.y) + q.x) + q.y;
}
#ifndef STAGE_ILP
#define STAGE_ILP 4
#endif
#define UNIT_W (8 * STAGE_ILP)    // columns of a work unit (round-4 form; the balanced form below cuts its own blocks)
#ifndef BORDER_ILP
#define BORDER_ILP 2            // strips that cross the image border: shorter trips keep the register budget of the hot path
#endif
// ---- interior rounds of the balanced form -----------------------------------------------------------------------------------------
// N samples of one lane, 8 columns apart, starting at column j0 + lj of the lane's row: positions px0 + 8 u c (8 c, 16 c, 24 c are exact
// doubles, so the sum rounds like start + j * step whenever that is exact), N gathers issued back to back, then the arithmetic; the LDS
// bytes go to wrow + 8 u -- immediate offsets of ds_write_b8, no address arithmetic per sample.
#ifndef STAGE_FRAC_FIRST
#define STAGE_FRAC_FIRST 1
#endif
template <int N>
__device__ __forceinline__ void stage_round(g_cu8 ubase, const uint32_t pw, const double c, const double sn, const double sxc, const double syc,
                                            const int jlane, uint8_t *wrow)
{
    uint32_t top[N];
    const double jd = (double)jlane;
    const double px0 = __builtin_fma(jd, c, sxc), py0 = __builtin_fma(jd, -sn, syc);     // start + j * step: the product is exact in double
#if STAGE_FRAC_FIRST
    // Round 6: the fractions are taken BEFORE the gathers are issued, so what waits for the loads are 2 N floats, not 2 N doubles -- the
    // positions were the kernel's widest live range (16 registers per round of four), and the kernel is bound by how many waves fit a SIMD
    // (profiles/r06_ab_occupancy.txt: six workgroups per CU instead of five: -3.7 %).
    float fa[N], fb[N];
#pragma unroll
    for (int u = 0; u < N; u++) {
        const double px = u ? px0 + (double)(8 * u) * c : px0, py = u ? py0 - (double)(8 * u) * sn : py0;
        const uint32_t off = ((uint32_t)__umul24((uint32_t)(int)py, pw) + (uint32_t)(int)px) << 1;
        fa[u] = (float)__builtin_amdgcn_fract(px); fb[u] = (float)__builtin_amdgcn_fract(py);
        top[u] = *(GAS const uint32_t *)(ubase + off);              // 2-byte-aligned dword gather, uniform base + 32-bit offset
    }
#pragma unroll
    for (int u = 0; u < N; u++) wrow[8 * u] = round_u8_pos(bilinear_pk(top[u], fa[u], fb[u]));
#else
    double px[N], py[N];
    px[0] = px0; py[0] = py0;
#pragma unroll
    for (int u = 1; u < N; u++) { px[u] = px[0] + (double)(8 * u) * c; py[u] = py[0] - (double)(8 * u) * sn; }
#pragma unroll
    for (int u = 0; u < N; u++) {
        const uint32_t off = ((uint32_t)__umul24((uint32_t)(int)py[u], pw) + (uint32_t)(int)px[u]) << 1;
        top[u] = *(GAS const uint32_t *)(ubase + off);              // 2-byte-aligned dword gather, uniform base + 32-bit offset
    }
#pragma unroll
    for (int u = 0; u < N; u++) {
        const float a = (float)__builtin_amdgcn_fract(px[u]), b = (float)__builtin_amdgcn_fract(py[u]);
        wrow[8 * u] = round_u8_pos(bilinear_pk(top[u], a, b));
    }
#endif
}

// (Round 6, measured and removed: issuing BOTH rounds of a 5..8-group unit before finishing the first -- the second round's gathers behind the
// first round's arithmetic -- needs 24 more live registers: 41 spills at 80 VGPRs, 24 at 96, some inside the rounds: describe 3.50 -> 6.25 /
// 5.80 ms on the 16-pair batch, profiles/r06_ab_stage_pipe.txt.)
// Balanced form (round 5).  A work unit is (strip of 8 rows) x (a run of 8-column GROUPS): the G8 = ceil(win / 8) groups of a row are cut
// into nb runs of floor / ceil(G8 / nb) groups -- at most 8 (64 columns) --, so that no unit is padded: the round-4 form cut 32-column
// blocks from the left, and a 140-px window cost five of them (160 columns), a 42-px one two (64).  A unit samples its groups in rounds of
// four (then 3, 2 or 1: a round is code of its own, stage_round<N>), its bookkeeping -- row origins, interior flag, LDS row, first column
// as a double -- is paid once per up to 8 samples of a lane instead of once per 4, and the group that hangs over the window's last column
// (win % 8 != 0) is a clamped round of its own, done by the unit that holds the row's last run.
template <int NW>                                              // NW waves share the strips (4: the workgroup; 1: one wave on its own)
__device__ __forceinline__ void stage_rows(const WinGeom &G, const float *sx_row, const float *sy_row,
                                           int r0, int nrows, uint8_t *dst, uint8_t *strip_in)
{
    // everything that is the same for the whole wave is pinned to SGPRs (the compiler cannot know that a keypoint read through a
    // ticket index, or threadIdx.x >> 6, is wave-uniform): the unit bookkeeping below then runs on the scalar unit, not on the VALU
    // that bounds this kernel
    const int win = __builtin_amdgcn_readfirstlane(G.win);
    r0 = __builtin_amdgcn_readfirstlane(r0); nrows = __builtin_amdgcn_readfirstlane(nrows);
    const int lane = threadIdx.x & 63, wv = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane >> 3, lj = lane & 7;
    const int strips = (nrows + 7) >> 3;
    const double c = (double)G.cos_dir, sn = (double)G.sin_dir;
    const int ncols1 = __builtin_amdgcn_readfirstlane(G.w) - 1, nrows1 = __builtin_amdgcn_readfirstlane(G.h) - 1;
    // Samples read the ROW-PAIR image (k_pair_rows): element (y, x) = pixel (y, x) | pixel (y + 1, x) << 8, so the four taps of a
    // bilinear sample are ONE dword at element (iy, ix) -- bytes t00, t10, t01, t11.  Its base is the same for the whole workgroup:
    // pinned to SGPRs so gathers use scalar-base + 32-bit-offset addressing.
    const uint64_t bp = (uint64_t)G.pair;
    // (readfirstlane returns int: widen through uint32_t, or a low half with bit 31 set sign-extends into the high half)
    g_cu8 ubase = (g_cu8)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(bp >> 32)) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)bp));
    const uint32_t pw = (uint32_t)(ncols1 + 1);                  // pitch of the pair image in elements
    // the cut of a row into runs of groups: full groups only (the partial last group is the tail round of the last run)
    const int gfull = win >> 3, tailc = win & 7;                 // win >= 21: gfull >= 2
    // runs of at most 8 groups.  (Measured, fixed 16-pair batch: shorter runs -- at most 4 groups, the round-4 unit size without its padding --
    // +2.5 %; a per-call choice of the run count by a cost model of the busiest wave +4 %: its scalar divisions cost more than the even
    // finish buys.)
    int nb = NW == 1 ? 1 : (gfull + 7) >> 3;                     // (the one-wave form stages windows of at most 64 px: one run per strip, known at compile time)
#if VFSMS_EXP & 16
    nb = (gfull + 3) >> 2;
#endif
    const int gbase = gfull / nb, grem = gfull - gbase * nb;     // run b holds gbase + (b < grem) groups and starts at b gbase + min(b, grem)
    const int total = strips * nb;
    // Which strips lie inside the image as a whole (all `win` columns)?  One lane per strip answers once for everybody (separable
    // extremes: row-origin extreme + column-step extreme, over the full width); units of such strips skip their own test.
    if (!G.upright) {
        const int tid = NW == 1 ? lane : (int)threadIdx.x;
        if (tid < strips) {
            const int ia = min(r0 + tid * 8, VFSMS_MAX_WIN - 1), ib = min(min(r0 + tid * 8 + 7, r0 + nrows - 1), VFSMS_MAX_WIN - 1);
            const double xa = (double)sx_row[ia], xb = (double)sx_row[ib], ya = (double)sy_row[ia], yb = (double)sy_row[ib];
            const double jb = (double)(win - 1);
            const double jxb = jb * c, jyb = -(jb * sn);
            const double xmin = fmin(xa, xb) + fmin(0.0, jxb), xmax = fmax(xa, xb) + fmax(0.0, jxb);
            const double ymin = fmin(ya, yb) + fmin(0.0, jyb), ymax = fmax(ya, yb) + fmax(0.0, jyb);
            strip_in[tid] = xmin >= 0.0 && ymin >= 0.0 && xmax < (double)(ncols1 - 2) && ymax < (double)nrows1;
        }
        if (NW == 1) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    }
    int ty = 0, cbi = wv;                                        // (strip, run) of the wave's current unit; stepped, not divided
    while (cbi >= nb) { cbi -= nb; ty++; }
#ifdef VFSMS_DESC_TIMING
    unsigned long long _s0 = clock64(), _uacc[2] = {0, 0}, _ucnt[2] = {0, 0};
#endif
    const int last_row = r0 + nrows - 1;                         // (<= win - 1 < VFSMS_MAX_WIN)
    const int liw = li * win;                                    // the lane's row inside a strip, as an LDS offset
    const int last_off = (nrows - 1) * win;
    // The row origins and the strip flag of a unit sit at the head of its dependency chain (LDS read -> f64 position -> address ->
    // gather): they are fetched ONE UNIT AHEAD, so the chain of a unit starts with values that are already in registers.
    float nsx = 0.f, nsy = 0.f; int nflag = 0;
    if (wv < total) {
        const int ic0 = min(r0 + ty * 8 + li, last_row);
        nsx = sx_row[ic0]; nsy = sy_row[ic0]; nflag = strip_in[ty];
    }
    for (int unit = wv; unit < total; unit += NW) {
        DT_UNIT_BEGIN;
        const int g0 = cbi * gbase + min(cbi, grem), gn = gbase + (cbi < grem ? 1 : 0);
        const bool last_run = cbi == nb - 1;
        const int cb0 = g0 * 8;
        const int cb1 = last_run ? win : cb0 + gn * 8;
        // a lane past the strip's last row repeats that row: same position, same value, same LDS byte
        const double sxc = (double)nsx, syc = (double)nsy;
        const int flag_cur = nflag;
        const int ty_cur = ty;
        uint8_t *drc = dst + min(ty * 8 * win + liw, last_off);
        cbi += NW;
        while (cbi >= nb) { cbi -= nb; ty++; }
        if (unit + NW < total) {
            const int icn = min(r0 + ty * 8 + li, last_row);
            nsx = sx_row[icn]; nsy = sy_row[icn]; nflag = strip_in[ty];
        }
        if (G.upright) {
            for (int j = cb0 + lj; j < cb1; j += 8) drc[j] = (uint8_t)win_sample_upright(G, min(r0 + ty_cur * 8 + li, last_row), j);
            continue;
        }
        // Unit-level interior test: x = origin(row) + j * c is separable, rows are monotone, so the extremes of the unit's samples are
        // (extreme row origin) + (extreme of j * c).  A unit inside the image (with the 2 px of slack the dword taps need) runs
        // without any per-sample bounds logic.
        bool unit_in = __builtin_amdgcn_readfirstlane(flag_cur) != 0;
        if (!unit_in) {
            const int ia = min(r0 + ty_cur * 8, VFSMS_MAX_WIN - 1), ib = min(min(r0 + ty_cur * 8 + 7, last_row), VFSMS_MAX_WIN - 1);
            const double xa = (double)sx_row[ia], xb = (double)sx_row[ib], ya = (double)sy_row[ia], yb = (double)sy_row[ib];
            const double ja = (double)cb0, jb = (double)(cb1 - 1);
            const double jxa = ja * c, jxb = jb * c, jya = -(ja * sn), jyb = -(jb * sn);
            const double xmin = fmin(xa, xb) + fmin(jxa, jxb), xmax = fmax(xa, xb) + fmax(jxa, jxb);
            const double ymin = fmin(ya, yb) + fmin(jya, jyb), ymax = fmax(ya, yb) + fmax(jya, jyb);
            unit_in = xmin >= 0.0 && ymin >= 0.0 && xmax < (double)(ncols1 - 2) && ymax < (double)nrows1;
        }
        if (unit_in) {
            // No predicates here: every lane gathers from inside the tested unit, the loads keep the scalar-base form and there is no
            // exec-mask bookkeeping around them.
            DT_TRIP(0);
            uint8_t *wrow = drc + cb0 + lj;
            int jl = cb0 + lj, left = gn;
            while (left >= 4) { stage_round<4>(ubase, pw, c, sn, sxc, syc, jl, wrow); jl += 32; wrow += 32; left -= 4; }
            if (left == 3) stage_round<3>(ubase, pw, c, sn, sxc, syc, jl, wrow);
            else if (left == 2) stage_round<2>(ubase, pw, c, sn, sxc, syc, jl, wrow);
            else if (left == 1) stage_round<1>(ubase, pw, c, sn, sxc, syc, jl, wrow);
            if (last_run && tailc) {
                // the group that hangs over the last column: lanes past it repeat that column (same position, same byte)
                const int jc = min(gfull * 8 + lj, win - 1);
                stage_round<1>(ubase, pw, c, sn, sxc, syc, jc, drc + jc);
            }
            DT_UNIT_END(0);
            continue;
        }
        // The unit crosses the image border.  No branch per sample: every lane gathers at clamped coordinates -- the bilinear taps
        // when (ix, iy) is interior, the nearest pixel clamp(cvRound(px), cvRound(py)) otherwise -- and selects at the end.
        DT_TRIP(2);
        for (int jb = cb0; jb < cb1; jb += 8 * BORDER_ILP) {
            double px[BORDER_ILP], py[BORDER_ILP];
            int jc[BORDER_ILP];
            uint32_t top[BORDER_ILP];                         // pair elements (cy, cx) and (cy, cx + 1) as ONE dword (round 6; two 16-bit gathers before)
            bool inside[BORDER_ILP];
#pragma unroll
            for (int u = 0; u < BORDER_ILP; u++) {
                jc[u] = min(jb + u * 8 + lj, cb1 - 1);
                px[u] = sxc + (double)jc[u] * c;
                py[u] = syc - (double)jc[u] * sn;
                const int ix = (int)px[u], iy = (int)py[u];                       // trunc == floor wherever `inside` holds
                inside[u] = px[u] >= 0.0 && py[u] >= 0.0 && ix < ncols1 && iy < nrows1;
                const int rx = min(max(cv_round_d(px[u]), 0), ncols1), ry = min(max(cv_round_d(py[u]), 0), nrows1);
                const int cx = inside[u] ? ix : rx, cy = inside[u] ? iy : ry;
                // An inside sample has ix + 1 <= w - 1: its dword is the four taps, as in the interior rounds.  An outside sample only uses
                // the low byte, pixel (cy, cx); at cx = w - 1 the dword runs into the next row's first element -- or, on the last row, into
                // the 8 bytes of slack behind the pair image (surf_roi_carve) -- whose bytes are never looked at.
                top[u] = *(GAS const uint32_t *)(ubase + (((uint32_t)__umul24((uint32_t)cy, pw) + (uint32_t)cx) << 1));
            }
#pragma unroll
            for (int u = 0; u < BORDER_ILP; u++) {
                const float a = (float)__builtin_amdgcn_fract(px[u]), b = (float)__builtin_amdgcn_fract(py[u]);
                // bytes t00, t10, t01, t11 (the high byte of an element is row min(cy + 1, h - 1)); the reference's operation order
                const float v = (float)(top[u] & 0xff) * (1.f - a) * (1.f - b) + (float)((top[u] >> 16) & 0xff) * a * (1.f - b) +
                                (float)((top[u] >> 8) & 0xff) * (1.f - a) * b + (float)(top[u] >> 24) * a * b;
                drc[jc[u]] = inside[u] ? (uint8_t)cv_round_f(v) : (uint8_t)(top[u] & 0xff);
            }
        }
        DT_UNIT_END(1);
    }
#ifdef VFSMS_DESC_TIMING
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_desc_unit_cycles[2], clock64() - _s0);
        atomicAdd(&g_desc_unit_cycles[0], _uacc[0]); atomicAdd(&g_desc_unit_cycles[1], _uacc[1]);
        atomicAdd(&g_desc_trips[0], _ucnt[0]); atomicAdd(&g_desc_trips[2], _ucnt[1]);
    }
#endif
}

__device__ __forceinline__ uint8_t sat_u8(float v)
{
    int iv = cv_round_f(v);
    return (uint8_t)(iv < 0 ? 0 : iv > 255 ? 255 : iv);
}

// computeResizeAreaTab per window size, built ONCE per context on the host (ctx_prepare_area_tab): record dx < 21 of window `win` = the run of
// source pixels [j0, j0 + n) of cell dx with alpha a0 for the first, af for the middle ones, al for the last (left partial / full / right
// partial); record 21 = { min n, max n, 1 / iscale^2, -, -, mode } with mode 0: float table, 1: integer scale, 2: scale == 2.
// The same records serve the rows.  For integer scales every alpha is 1: the float sums are the exact integer sums of the fast paths.
struct AreaRec { int j0, n; float a0, af, al; int mode, r1, r2; };
#define AREA_RECS 22

// buf[dx] of one source row: S[j0] * a0 + S[j0 + 1] * af + ... + S[j0 + n - 1] * al, left to right.  The trip count is the same for every
// lane (nmax); steps past a cell's own run add S * 0 (x + 0 == x: the sum is never -0), so there is no per-lane loop bound.
__device__ __forceinline__ float area_row_tab(const uint8_t *S, const AreaRec &c, const int nmin, const int nmax)
{
    const uint8_t *p = S + c.j0;
    float buf = (float)p[0] * c.a0;
    int t = 1;
    for (; t < nmin - 1; t++) buf += (float)p[t] * c.af;
    for (; t < nmax; t++) buf += (float)p[t] * (t < c.n - 1 ? c.af : (t == c.n - 1 ? c.al : 0.f));
    return buf;
}
// The same sum with the weights of the taps behind the common part -- taps max(1, nmin - 1) .. nmax - 1: af, al or 0 depending on the cell's own
// count -- formed ONCE per cell (area_tail_weights) instead of by two compares and two selects per tap and row: the products and their order
// are unchanged.  At most AREA_TAIL taps lie there (cell counts of a window differ by at most 2); callers fall back to area_row_tab otherwise.
#define AREA_TAIL 4
__device__ __forceinline__ void area_tail_weights(const AreaRec &c, const int nmin, float (&w)[AREA_TAIL])
{
    const int t0 = max(nmin - 1, 1);
#pragma unroll
    for (int i = 0; i < AREA_TAIL; i++) w[i] = t0 + i < c.n - 1 ? c.af : (t0 + i == c.n - 1 ? c.al : 0.f);
}
__device__ __forceinline__ float area_row_tab_w(const uint8_t *S, const AreaRec &c, const int nmin, const int nmax, const float (&w)[AREA_TAIL])
{
    const uint8_t *p = S + c.j0;
    float buf = (float)p[0] * c.a0;
    int t = 1;
    for (; t < nmin - 1; t++) buf += (float)p[t] * c.af;
#pragma unroll
    for (int i = 0; i < AREA_TAIL; i++)
        if (t + i < nmax) buf += (float)p[t + i] * w[i];
    return buf;
}
// one output pixel from the row sums rb[0], rb[21], ... of its n source rows (sum = beta * buf, then sum += beta * buf)
__device__ __forceinline__ uint8_t area_col_tab(const float *rb, const AreaRec &c, const int mode, const float inv_area)
{
    float sum = c.a0 * rb[0];
    for (int t = 1; t < c.n; t++) sum += (t < c.n - 1 ? c.af : c.al) * rb[t * 21];
    if (mode == 2) return (uint8_t)(((int)sum + 2) >> 2);
    if (mode == 1) return sat_u8(sum * inv_area);
    return sat_u8(sum);
}

// band >= 0: only row `band` of the 21 x 21 patch (tickets of the largest windows are split by output row, see k_desc_plan)
__device__ void describe_one(const RoiDev &R, const SurfTables *T, const AreaRec *area_tab, const DescRec &rec, int extended, int upright, const int band)
{
    DT_START;
    const int k = rec.k;
    __shared__ float sx_row[VFSMS_MAX_WIN], sy_row[VFSMS_MAX_WIN];
    __shared__ uint8_t PATCH[21][21 + 3];
    __shared__ AreaRec REC[AREA_RECS];                     // computeResizeAreaTab of this window size: same records for x and y
    __shared__ uint8_t WINBUF[DESC_WBUF];
    __shared__ float rowsum[40 * 21];                      // buf[dx] of up to 40 source rows
    __shared__ uint8_t STRIP_IN[VFSMS_MAX_WIN / 8 + 8];    // per 8-row strip of a staging call: inside the image as a whole?
    WinGeom G;
    G.win = max(21, min(rec.win, VFSMS_MAX_WIN));          // (the record is wave-uniform: scalar registers, and so is what derives from it)
    G.h = R.h; G.w = R.w; G.stride = R.stride; G.img = (g_cu8)R.img; G.pair = (g_cu8)R.pair;
    G.upright = upright; G.usx = 0; G.usy = 0; G.sin_dir = rec.sin_dir; G.cos_dir = rec.cos_dir;
    const int win = G.win;
    const int dsz = 21;
    if (threadIdx.x < AREA_RECS * 8) ((int *)REC)[threadIdx.x] = ((const int *)(area_tab + (size_t)win * AREA_RECS))[threadIdx.x];
    __syncthreads();
    const int nmin = __builtin_amdgcn_readfirstlane(REC[21].j0), nmax = __builtin_amdgcn_readfirstlane(REC[21].n);
    const int mode = __builtin_amdgcn_readfirstlane(REC[21].mode);
    const float inv_area = REC[21].a0;
    if (!upright) {
        // Row origins are running float sums in the reference (start_x += sin_dir per row): inherently sequential, so one lane of wave 0
        // walks x while one lane of wave 1 walks y (both chains in lanes 0 / 1 of ONE wave -- half the issue slots -- measured 0.7 %
        // SLOWER: the prologue is latency, not issue).  sin / cos of the orientation come with the keypoint's record (k_desc_recs).
        if (threadIdx.x == 0 || threadIdx.x == 64) {
            const float sin_dir = rec.sin_dir, cos_dir = rec.cos_dir;
            const float win_offset = -(float)(win - 1) / 2;
            // a band ticket only needs the origins up to the last source row of its band
            const int need = band >= 0 ? min(win, REC[band].j0 + REC[band].n) : win;
            if (threadIdx.x == 0) origin_chain(sx_row, rec.x + win_offset * cos_dir + win_offset * sin_dir, sin_dir, need);
            else origin_chain(sy_row, rec.y - win_offset * sin_dir + win_offset * cos_dir, cos_dir, need);
        }
    } else {
        const float win_offset = -(float)(win - 1) / 2;
        G.usx = cv_round_f(rec.x + win_offset);
        G.usy = cv_round_f(rec.y - win_offset);
    }
    __syncthreads();
    DT_MARK(0);

    // The rotated window is staged through LDS so that every bilinear sample is produced exactly once: the whole
    // window when it fits (win <= 128), otherwise as many rows of output cells as fit.  INTER_AREA then runs from LDS as two
    // table-driven passes in cv::resize's accumulation order: buf[dx] of every staged source row (one (row, cell) per lane, uniform
    // trip counts), then the output pixels from those row sums -- in chunks of whole output rows whose source rows fit `rowsum`.
    // rows_lo .. rows_hi of WINBUF row index 0 hold window rows st_lo ..; reduce the output rows [dyA, dyB_end)
    const bool tail_ok = nmax - max(nmin - 1, 1) <= AREA_TAIL;     // (always, for the tables computeResizeAreaTab makes; wave-uniform)
    auto reduce_rows = [&](const int st_lo, int dyA, const int dy_stop) {
        while (dyA < dy_stop) {
            const int lo = REC[dyA].j0;
            int dyB = dyA + 1;
            while (dyB < dy_stop && REC[dyB].j0 + REC[dyB].n - lo <= 40) dyB++;
            const int nr = REC[dyB - 1].j0 + REC[dyB - 1].n - lo;
            for (int e = threadIdx.x; e < nr * 21; e += 256) {
                const int r = (int)(((uint32_t)e * 3121u) >> 16), dx = e - 21 * r;      // e / 21, exact for e < 43690
                if (tail_ok) {
                    const AreaRec cx = REC[dx];
                    float w[AREA_TAIL];
                    area_tail_weights(cx, nmin, w);
                    rowsum[e] = area_row_tab_w(WINBUF + (lo - st_lo + r) * win, cx, nmin, nmax, w);
                } else
                    rowsum[e] = area_row_tab(WINBUF + (lo - st_lo + r) * win, REC[dx], nmin, nmax);
            }
            __syncthreads();
            for (int o = threadIdx.x; o < (dyB - dyA) * 21; o += 256) {
                const int dyl = (int)(((uint32_t)o * 3121u) >> 16), dx = o - 21 * dyl;
                const AreaRec ry = REC[dyA + dyl];
                PATCH[dyA + dyl][dx] = area_col_tab(rowsum + (ry.j0 - lo) * 21 + dx, ry, mode, inv_area);
            }
            __syncthreads();
            dyA = dyB;
        }
    };
    if (win * win <= DESC_WBUF) {
        stage_rows<4>(G, sx_row, sy_row, 0, win, WINBUF, STRIP_IN);
        __syncthreads();
        DT_MARK(1);
        reduce_rows(0, 0, dsz);
        DT_MARK(2);
    } else {
        const int dy_end = band >= 0 ? band + 1 : dsz;
        const int crows = max(DESC_WBUF / win, 1);                 // window rows the LDS buffer holds (>= 22 for win <= 739)
        for (int dy = band >= 0 ? band : 0; dy < dy_end;) {
            const int rlo = REC[dy].j0, nrows = REC[dy].n;         // <= scale + 2 <= 39
            if (nrows <= crows) {
                // Super-band: the buffer is filled with the source rows of as many consecutive output rows as fit.  A band on
                // its own is only scale + 2 (8-38) rows tall -- one or two 8-row strips, a third of their lanes idle, and every row
                // shared by two bands sampled twice; a full buffer is sampled in whole strips and its shared rows once.
                int d_stop = dy + 1;
                while (d_stop < dy_end && REC[d_stop].j0 + REC[d_stop].n - rlo <= crows) d_stop++;
                const int end = REC[d_stop - 1].j0 + REC[d_stop - 1].n - 1;
                stage_rows<4>(G, sx_row, sy_row, rlo, end - rlo + 1, WINBUF, STRIP_IN);
                __syncthreads();
                DT_MARK(3);
                reduce_rows(rlo, dy, d_stop);
                DT_MARK(4);
                dy = d_stop;
            } else {
                // a band taller than the buffer (win > 409): staged in chunks of its own, row sums collected over the chunks
                for (int c0 = 0; c0 < nrows; c0 += crows) {
                    const int cn = min(crows, nrows - c0);
                    stage_rows<4>(G, sx_row, sy_row, rlo + c0, cn, WINBUF, STRIP_IN);
                    __syncthreads();
                    DT_MARK(3);
                    for (int e = threadIdx.x; e < cn * 21; e += 256) {
                        const int r = (int)(((uint32_t)e * 3121u) >> 16), dx = e - 21 * r;
                        rowsum[c0 * 21 + e] = area_row_tab(WINBUF + r * win, REC[dx], nmin, nmax);
                    }
                    __syncthreads();
                }
                if (threadIdx.x < dsz) PATCH[dy][threadIdx.x] = area_col_tab(rowsum + threadIdx.x, REC[dy], mode, inv_area);
                __syncthreads();
                DT_MARK(4);
                dy++;
            }
        }
    }
    __syncthreads();
    // hand the 21 x 21 patch to k_desc_tail (gradients, cell sums, normalisation run there, 16 keypoints per workgroup)
    if (band >= 0) {
        if (threadIdx.x < 21) R.patch[(size_t)k * VFSMS_PATCH_ROW + band * 21 + threadIdx.x] = PATCH[band][threadIdx.x];
    } else {
        for (int o = threadIdx.x; o < 441; o += 256) R.patch[(size_t)k * VFSMS_PATCH_ROW + o] = PATCH[o / 21][o % 21];
    }
    DT_MARK(5);
}

// ---------------------------------------------------------------------------------------------------
// Persistent scheduling for the descriptor kernel.  The number of keypoints of each ROI only exists on the device
// (no host sync inside a batch), so instead of launching a capacity-sized grid -- mostly workgroups that find
// nothing to do, each still paying a 30 KB LDS allocation -- 5 resident workgroups per CU draw (ROI, keypoint)
// tickets from one atomic counter over the concatenation of all ROIs' keypoint lists (prefix sums of the
// device-side counts, rebuilt per workgroup in LDS).  Window cost varies 100x between keypoints, so dynamic
// tickets matter: static striding measured 45 % slower, 4-ticket chunks 20 % slower.
// ---------------------------------------------------------------------------------------------------
#define DESC_NCLS 4                  // window-size classes, largest first: win > 256, > 128, > 64, rest

// Tickets are drawn class by class, largest descriptor windows first: one window of 700 px costs as much as 300 windows of 40 px,
// and in response order such a window could be drawn last and leave the whole chip waiting for a single workgroup.  k_desc_order
// builds, per ROI, the list of surviving keypoints grouped by class (a counting sort by one workgroup); the order inside a class
// is irrelevant, every keypoint is described independently into its own patch row.
__device__ __forceinline__ int desc_class(float size)
{
    const float s = size * 1.2f / 9.0f;
    const int win = min((int)((20 + 1) * s), VFSMS_MAX_WIN);
    return win > 256 ? 0 : win > 128 ? 1 : win > 64 ? 2 : 3;
}

__global__ __launch_bounds__(1024) void k_desc_order(const RoiDev *rois)
{
    const RoiDev &R = rois[blockIdx.x];
    const int n = min(R.counters[0], R.cap);
    __shared__ int cnt[DESC_NCLS], base[DESC_NCLS], cur[DESC_NCLS];
    if (threadIdx.x < DESC_NCLS) { cnt[threadIdx.x] = 0; cur[threadIdx.x] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int k0 = 0; k0 < n; k0 += 1024) {
        const int k = k0 + threadIdx.x;
        const float size = k < n ? R.kps[k].size : -1.f;
        const int cls = size > 0 ? desc_class(size) : -1;
#pragma unroll
        for (int c = 0; c < DESC_NCLS; c++) {
            const unsigned long long m = __ballot(cls == c);
            if (m && lane == __ffsll((long long)m) - 1) atomicAdd(&cnt[c], __popcll(m));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int c = 0; c < DESC_NCLS; c++) { base[c] = acc; acc += cnt[c]; R.counters[12 + c] = cnt[c]; }
    }
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += 1024) {
        const int k = k0 + threadIdx.x;
        const float size = k < n ? R.kps[k].size : -1.f;
        const int cls = size > 0 ? desc_class(size) : -1;
#pragma unroll
        for (int c = 0; c < DESC_NCLS; c++) {
            const unsigned long long m = __ballot(cls == c);
            if (!m) continue;
            const int leader = __ffsll((long long)m) - 1;
            int start = 0;
            if (lane == leader) start = atomicAdd(&cur[c], __popcll(m));
            start = __shfl(start, leader, 64);
            if (cls == c) R.order[base[c] + start + __popcll(m & ((1ull << lane) - 1))] = k;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Work list of the descriptor kernels (round 5).  The ticket counter is sharded 8 ways (one head per XCD, 256 B apart: a single
// device-scope word saturates near 90 returning atomics per microsecond) and the heads are XCD-AFFINE: head q owns the ROIs q, q + 8, ...
// and serves them one after the other, so the 160 workgroups of an XCD sample one or two pair images at a time -- 1.7 MB each, inside the
// XCD's 4 MB of L2 -- instead of all 80 of a launch (round 4's one class-major order over all ROIs: every XCD drew tickets of every ROI,
// PMC FETCH_SIZE 0.94 GB per launch of 40 pairs for k_describe and 0.61 GB for k_describe_small, now 0.11 GB each).  A workgroup starts
// on the head of its XCD (blockIdx % 8: workgroups are dealt to the XCDs round-robin) and moves on to the next head when its own runs dry.
// What a ticket leads to is ONE 32-byte record (DescRec) at a position that follows from the ticket alone: k_desc_plan lays the heads'
// slices out, k_desc_recs fills them.
// ---------------------------------------------------------------------------------------------------
#define DESC_HEADS DESC_PLAN_HEADS
#define DESC_HEAD_STRIDE 64          // ints between heads

// one workgroup: the layout of the record arrays from the class counts of all ROIs (counters[12..15], k_desc_order)
__global__ __launch_bounds__(256) void k_desc_plan(const RoiDev *rois, int nrois, DescPlan *plan, int big_grid)
{
    __shared__ int c[VFSMS_MAX_ROIS][4];
    for (int e = threadIdx.x; e < nrois * 4; e += 256) c[e >> 2][e & 3] = rois[e >> 2].counters[12 + (e & 3)];
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0, total = 0;
        for (int q = 0; q < DESC_HEADS; q++) {
            plan->big_start[q] = acc;
            for (int r = q; r < nrois; r += DESC_HEADS) { plan->seg_base[r][0] = acc; acc += c[r][0]; }
            plan->big_n0[q] = acc - plan->big_start[q];
            for (int r = q; r < nrois; r += DESC_HEADS) {
                plan->seg_base[r][1] = acc; acc += c[r][1];
                plan->seg_base[r][2] = acc; acc += c[r][2];
            }
            plan->big_tickets[q] = acc - plan->big_start[q];
        }
        total = acc;
        // Small batches are bounded by their few largest windows: those (class 0, win > 256) are then drawn as 21 tickets each, one per
        // output row of the patch.  Large batches keep one ticket per keypoint (the split repeats the row-origin prologue 21 times).
        // "Small" = fewer than ~48 keypoints OF THIS KERNEL (classes 0-2: `total` does not count class 3, which k_describe_small serves
        // from its own grid) per resident workgroup of this kernel -- since round 5; round 4 counted all four classes against the same
        // bound, i.e. switched at a third of the batch size.  With six workgroups per CU the switch sits at ~74 k keypoints of classes
        // 0-2 (~24 ROIs of 409 x 2048).
        const int split = total < big_grid * 48 ? 21 : 1;
        plan->split = split;
        for (int q = 0; q < DESC_HEADS; q++) plan->big_tickets[q] += (split - 1) * plan->big_n0[q];
        acc = 0;
        for (int q = 0; q < DESC_HEADS; q++) {
            plan->small_start[q] = acc;
            for (int r = q; r < nrois; r += DESC_HEADS) { plan->seg_base[r][3] = acc; acc += c[r][3]; }
            plan->small_tickets[q] = acc - plan->small_start[q];
        }
    }
}

// One thread per surviving keypoint (position p of the ROI's class-major order list): its record, with sin / cos of the descriptor window's
// rotation (std::sin / std::cos on float in the reference).  det_sincos (detmath.h) is the explicit double-precision algorithm the oracle
// evaluates too, so both sides round to the same float; it agrees with a correctly rounded sinf / cosf except within ~2^-29 ulp of a
// rounding boundary.
__global__ __launch_bounds__(256) void k_desc_recs(const RoiDev *rois, const DescPlan *plan, DescRec *big, DescRec *small, int upright)
{
    const int roi = blockIdx.y;
    const RoiDev &R = rois[roi];
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int c0 = R.counters[12], c1 = R.counters[13], c2 = R.counters[14], c3 = R.counters[15];
    if (p >= c0 + c1 + c2 + c3) return;
    const int cls = p < c0 ? 0 : p < c0 + c1 ? 1 : p < c0 + c1 + c2 ? 2 : 3;
    const int idx = p - (cls > 0 ? c0 : 0) - (cls > 1 ? c1 : 0) - (cls > 2 ? c2 : 0);
    const int k = R.order[p];
    const vfsms_keypoint kp = R.kps[k];
    DescRec rec;
    rec.roi = roi; rec.k = k; rec.pad = 0; rec.x = kp.x; rec.y = kp.y;
    rec.win = (int)((20 + 1) * (kp.size * 1.2f / 9.0f));
    rec.sin_dir = 0.f; rec.cos_dir = 0.f;
    if (!upright) {
        const float dir = kp.angle * (float)(3.1415926535897932384626433832795 / 180);
        double sd, cd;
        det_sincos((double)dir, &sd, &cd);
        rec.sin_dir = -(float)sd;
        rec.cos_dir = (float)cd;
    }
    (cls < 3 ? big : small)[plan->seg_base[roi][cls] + idx] = rec;
}

__global__ __launch_bounds__(ORI_KP * 128, 8) void k_orientation(const RoiDev *rois, const SurfTables *T, int upright)
{
    const RoiDev &R = rois[blockIdx.y];
    const int n = min(R.counters[0], R.cap);
    const int k0 = blockIdx.x * ORI_KP;
    if (k0 >= n) return;
    orientation_block(R, T, k0, n, upright);
}

// ---------------------------------------------------------------------------------------------------
// Windows of <= 64 px (class 3: two thirds of the keypoints, a quarter of the samples) are described by ONE WAVE each: a 42 px
// window is 7 samples per thread of a 256-thread workgroup, so the workgroup form spends its time in the five barriers, the
// two-barrier ticket draw and the single-lane row-origin chains.  Here the four waves of a workgroup run independently -- own ticket
// (lane 0 draws, the wave shares it by shuffle), own 64 x 64 LDS window, no workgroup barrier after the prefix table is built -- with
// the sampling and INTER_AREA arithmetic of describe_one (stage_rows<1>, the same cell code).
// ---------------------------------------------------------------------------------------------------
#define DESC_SMALL_WIN 64
struct SmallLds { uint8_t win[DESC_SMALL_WIN * DESC_SMALL_WIN + 64]; float sx[DESC_SMALL_WIN + 8], sy[DESC_SMALL_WIN + 8]; AreaRec rec[AREA_RECS]; int rec_win; uint8_t strip_in[DESC_SMALL_WIN / 8 + 8]; };

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ void describe_small(const RoiDev &R, const AreaRec *area_tab, const DescRec &rec, int upright, SmallLds &L)
{
    const int lane = threadIdx.x & 63;
    const int k = rec.k;
    WinGeom G;
    G.win = min(rec.win, DESC_SMALL_WIN);                      // (class 3 means <= 64 already; wave-uniform, scalar)
    G.h = R.h; G.w = R.w; G.stride = R.stride; G.img = (g_cu8)R.img; G.pair = (g_cu8)R.pair;
    G.upright = upright; G.usx = 0; G.usy = 0; G.sin_dir = rec.sin_dir; G.cos_dir = rec.cos_dir;
    const int win = G.win;
    const int dsz = 21;
    // computeResizeAreaTab of this window size from the host-built table (kept while the wave's next keypoint has the same window)
    if (L.rec_win != win) {
        wave_sync_lds();
        const int *src = (const int *)(area_tab + (size_t)max(win, dsz) * AREA_RECS);
        for (int e = lane; e < AREA_RECS * 8; e += 64) ((int *)L.rec)[e] = src[e];
        if (lane == 0) L.rec_win = win;
    }
    if (!upright) {
        const float sin_dir = rec.sin_dir, cos_dir = rec.cos_dir;
        if (lane < 2) {                                    // running float sums of the reference: lane 0 walks x, lane 1 walks y, in ONE loop
            const float win_offset = -(float)(win - 1) / 2;
            const float sx0 = rec.x + win_offset * cos_dir + win_offset * sin_dir, sy0 = rec.y - win_offset * sin_dir + win_offset * cos_dir;
            origin_chain(lane == 0 ? L.sx : L.sy, lane == 0 ? sx0 : sy0, lane == 0 ? sin_dir : cos_dir, win);
        }
    } else {
        const float win_offset = -(float)(win - 1) / 2;
        G.usx = cv_round_f(rec.x + win_offset);
        G.usy = cv_round_f(rec.y - win_offset);
    }
    wave_sync_lds();
    stage_rows<1>(G, L.sx, L.sy, 0, win, L.win, L.strip_in);
    wave_sync_lds();
    const int nmin = __builtin_amdgcn_readfirstlane(L.rec[21].j0), nmax = __builtin_amdgcn_readfirstlane(L.rec[21].n);
    const int mode = __builtin_amdgcn_readfirstlane(L.rec[21].mode);
    const float inv_area = L.rec[21].a0;
    const bool tail_ok = nmax - max(nmin - 1, 1) <= AREA_TAIL;
    uint8_t *prow = R.patch + (size_t)k * VFSMS_PATCH_ROW;
    // one output pixel per lane: sum over its source rows of beta * (row sum of its cell), uniform trip counts (surplus rows / pixels
    // weigh 0; they read at most nmax bytes past the window, inside L.win's padding or the row origins behind it)
    for (int o = lane; o < dsz * dsz; o += 64) {
        const int dy = (int)(((uint32_t)o * 3121u) >> 16), dx = o - dsz * dy;
        const AreaRec ry = L.rec[dy], rx = L.rec[dx];
        const uint8_t *S = L.win + ry.j0 * win;
        float sum;
        if (tail_ok) {                                     // the cell's tail weights once for all of its rows
            float w[AREA_TAIL];
            area_tail_weights(rx, nmin, w);
            sum = ry.a0 * area_row_tab_w(S, rx, nmin, nmax, w);
            for (int ty = 1; ty < nmax; ty++) {
                const float beta = ty < ry.n - 1 ? ry.af : (ty == ry.n - 1 ? ry.al : 0.f);
                sum += beta * area_row_tab_w(S + ty * win, rx, nmin, nmax, w);
            }
        } else {
            sum = ry.a0 * area_row_tab(S, rx, nmin, nmax);
            for (int ty = 1; ty < nmax; ty++) {
                const float beta = ty < ry.n - 1 ? ry.af : (ty == ry.n - 1 ? ry.al : 0.f);
                sum += beta * area_row_tab(S + ty * win, rx, nmin, nmax);
            }
        }
        uint8_t outv;
        if (mode == 2) outv = (uint8_t)(((int)sum + 2) >> 2);
        else if (mode == 1) outv = sat_u8(sum * inv_area);
        else outv = sat_u8(sum);
        prow[o] = outv;                                    // the 21 x 21 patch for k_desc_tail
    }
    wave_sync_lds();                                       // the next keypoint of this wave overwrites L
}

#ifndef DESC_SMALL_WGS
#define DESC_SMALL_WGS 7
#endif
// a wave-uniform record through the scalar unit: one s_load_dwordx8
__device__ __forceinline__ DescRec load_rec_uniform(const DescRec *p)
{
    const int *q = (const int *)p;
    DescRec r;
    r.roi = __builtin_amdgcn_readfirstlane(q[0]); r.k = __builtin_amdgcn_readfirstlane(q[1]);
    r.win = __builtin_amdgcn_readfirstlane(q[2]); r.pad = 0;
    r.sin_dir = __int_as_float(__builtin_amdgcn_readfirstlane(q[4])); r.cos_dir = __int_as_float(__builtin_amdgcn_readfirstlane(q[5]));
    r.x = __int_as_float(__builtin_amdgcn_readfirstlane(q[6])); r.y = __int_as_float(__builtin_amdgcn_readfirstlane(q[7]));
    return r;
}
__global__ __launch_bounds__(256, DESC_SMALL_WGS) void k_describe_small(const RoiDev *rois, const DescPlan *plan, const DescRec *recs, int *counter,
                                                                 const AreaRec *area_tab, int upright)
{
    __shared__ SmallLds L[4];
    if (threadIdx.x < 4) L[threadIdx.x].rec_win = -1;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int head = 0;
    for (;;) {
        int slot = -1;                                       // record index, lane 0's
        if (lane == 0)
            while (head < DESC_HEADS) {
                const int q = (blockIdx.x + head) & (DESC_HEADS - 1);
                const int n = plan->small_tickets[q];
                const int t = n > 0 ? atomicAdd(counter + q * DESC_HEAD_STRIDE, 1) : 0;
                if (t < n) { slot = plan->small_start[q] + t; break; }
                head++;                                      // this head is exhausted (it stays exhausted): steal from the next
            }
        slot = __builtin_amdgcn_readfirstlane(slot);         // lane 0's ticket, as an SGPR: the records below are scalar loads
        head = __builtin_amdgcn_readfirstlane(head);
        if (slot < 0) break;
        const DescRec rec = load_rec_uniform(recs + slot);
        describe_small(rois[rec.roi], area_tab, rec, upright, L[wave]);
    }
}

// Row-pair image for the descriptor sampling (see stage_rows): element (y, x) = pixel (y, x) | pixel (min(y + 1, h - 1), x) << 8.
// One thread per four pixels; grid (ceil(w / 1024), h, rois).
__global__ __launch_bounds__(256) void k_pair_rows(const RoiDev *rois)
{
    const RoiDev &R = rois[blockIdx.z];
    const int y = blockIdx.y, x = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (y >= R.h || x >= R.w) return;
    g_cu8 r0 = (g_cu8)R.img + (size_t)y * R.stride + x, r1 = (g_cu8)R.img + (size_t)min(y + 1, R.h - 1) * R.stride + x;
    GAS uint16_t *dst = (GAS uint16_t *)R.pair + (size_t)y * R.w + x;
    if (x + 4 <= R.w) {
        const uint32_t a = *(GAS const u32u *)r0, b = *(GAS const u32u *)r1;
#pragma unroll
        for (int k = 0; k < 4; k++) dst[k] = (uint16_t)(((a >> (8 * k)) & 0xff) | (((b >> (8 * k)) & 0xff) << 8));
    } else {
        for (int k = 0; x + k < R.w; k++) dst[k] = (uint16_t)(r0[k] | (r1[k] << 8));
    }
}

// Round 6: SIX workgroups per CU (80 VGPRs instead of 96: a dozen cold spills at kernel entry).  The kernel is bound by how many waves
// wait for gathers at a time, not by what it issues: 5 -> 6 workgroups -3.7 % (describe 3.70 -> 3.56 ms on the 16-pair batch), 7 (72
// VGPRs, 11 KB window buffer) no further gain; k_describe_small 6 -> 7: -1.5 % (profiles/r06_ab_occupancy.txt).
#ifndef DESC_WGS
#define DESC_WGS 6
#endif
__global__ __launch_bounds__(256, DESC_WGS) void k_describe(const RoiDev *rois, const DescPlan *plan, const DescRec *recs, int *counter, const SurfTables *T,
                                                  const AreaRec *area_tab, int extended, int upright)
{
    __shared__ int s_slot, s_band;
    int head = 0;                                            // thread 0's
#ifdef VFSMS_DESC_TIMING
    unsigned long long tq = clock64();
#endif
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int slot = -1, band = -1;
            const int split = plan->split;
            while (head < DESC_HEADS) {
                const int q = (blockIdx.x + head) & (DESC_HEADS - 1);
                const int n = plan->big_tickets[q];
                const int t = n > 0 ? atomicAdd(counter + q * DESC_HEAD_STRIDE, 1) : 0;
                if (t < n) {
                    // the head's class-0 records come first; with split > 1 each of them is `split` tickets, one per output row
                    const int n0 = plan->big_n0[q];
                    if (t < n0 * split) { slot = plan->big_start[q] + t / split; band = split > 1 ? t % split : -1; }
                    else slot = plan->big_start[q] + n0 + (t - n0 * split);
                    break;
                }
                head++;                                      // this head is exhausted (it stays exhausted): steal from the next
            }
            s_slot = slot; s_band = band;
        }
        __syncthreads();
        const int slot = __builtin_amdgcn_readfirstlane(s_slot), band = __builtin_amdgcn_readfirstlane(s_band);
        if (slot < 0) break;
#ifdef VFSMS_DESC_TIMING
        if (threadIdx.x == 0) atomicAdd(&g_desc_cycles[6], clock64() - tq);
#endif
        // the record and the ROI it names are the same for the whole workgroup: scalar loads, and everything derived from them stays off the VALU
        const DescRec rec = load_rec_uniform(recs + slot);
        describe_one(rois[rec.roi], T, area_tab, rec, extended, upright, band);
#ifdef VFSMS_DESC_TIMING
        tq = clock64();
#endif
    }
}

// ---------------------------------------------------------------------------------------------------
// deletion of size<=0 keypoints preserving order (SURF_Impl::detectAndCompute tail): single-workgroup
// scan per ROI + scatter.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_keep_scan(const RoiDev *rois)
{
    const RoiDev &R = rois[blockIdx.x];
    const int n = min(R.counters[0], R.cap);
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        int idx = base + threadIdx.x;
        int keep = (idx < n) ? (R.kps[idx].size > 0 ? 1 : 0) : 0;
        int incl = wave_incl_scan(keep);
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int off = carry;
        for (int k = 0; k < wid; k++) off += wsum[k];
        if (idx < n) R.keep_pos[idx] = keep ? (off + incl - 1) : -1;
        __syncthreads();
        if (threadIdx.x == 1023) carry = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) R.counters[1] = carry;
}


// Descriptor tail, 16 keypoints per workgroup, 16 threads (one per 5 x 5 cell) per keypoint: Gaussian-weighted
// gradients of the 21 x 21 patch, per-cell sums in raster order, L2 normalisation with the double accumulator of
// the reference, and the store straight into the COMPACTED descriptor / keypoint arrays (keep_pos from k_keep_scan).
__global__ __launch_bounds__(256) void k_desc_tail(const RoiDev *rois, const SurfTables *T, int extended)
{
    const RoiDev &R = rois[blockIdx.y];
    const int n = min(R.counters[0], R.cap);
    const int k0 = blockIdx.x * 16;
    if (k0 >= n) return;
    const int dsize = extended ? 128 : 64;
    __shared__ uint32_t P32[16][VFSMS_PATCH_ROW / 4];
    __shared__ float vec[16][128];
    __shared__ float scl[16];
    __shared__ float DWs[400];                               // the 20 x 20 Gaussian weights: 25 taps per thread come from LDS, not through the TA
    for (int i = threadIdx.x; i < 400; i += 256) DWs[i] = T->DW[i];
    {
        const uint32_t __attribute__((address_space(1))) *src =
            (const uint32_t __attribute__((address_space(1))) *)(R.patch + (size_t)k0 * VFSMS_PATCH_ROW);
        const int rows = min(16, n - k0);
        for (int i = threadIdx.x; i < rows * (VFSMS_PATCH_ROW / 4); i += 256) (&P32[0][0])[i] = src[i];
    }
    __syncthreads();
    const int kk = threadIdx.x >> 4, c = threadIdx.x & 15, k = k0 + kk;
    const int pos = k < n ? R.keep_pos[k] : -1;
    if (pos >= 0) {
        const uint8_t *P = (const uint8_t *)&P32[kk][0];
        const int ci = c >> 2, cj = c & 3;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int pb[6][6];                                       // the cell's 6 x 6 pixels, read once (each is a tap of up to four gradients)
#pragma unroll
        for (int yy = 0; yy < 6; yy++)
#pragma unroll
            for (int xx = 0; xx < 6; xx++) pb[yy][xx] = P[(ci * 5 + yy) * 21 + cj * 5 + xx];
#pragma unroll
        for (int yy = 0; yy < 5; yy++)
#pragma unroll
            for (int xx = 0; xx < 5; xx++) {
                const float dw = DWs[(ci * 5 + yy) * 20 + cj * 5 + xx];
                const int p00 = pb[yy][xx], p01 = pb[yy][xx + 1], p10 = pb[yy + 1][xx], p11 = pb[yy + 1][xx + 1];
                const float tx = (float)(p01 - p00 + p11 - p10) * dw;
                const float ty = (float)(p10 - p00 + p11 - p01) * dw;
                if (extended) {
                    if (ty >= 0) { v[0] += tx; v[1] += fabsf(tx); } else { v[2] += tx; v[3] += fabsf(tx); }
                    if (tx >= 0) { v[4] += ty; v[5] += fabsf(ty); } else { v[6] += ty; v[7] += fabsf(ty); }
                } else {
                    v[0] += tx; v[1] += ty; v[2] += fabsf(tx); v[3] += fabsf(ty);
                }
            }
        if (extended) {
#pragma unroll
            for (int q = 0; q < 8; q++) vec[kk][c * 8 + q] = v[q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) vec[kk][c * 4 + q] = v[q];
        }
    }
    __syncthreads();
    if (pos >= 0 && c == 0) {
        double square_mag = 0;
        for (int q = 0; q < dsize; q++) square_mag += (double)(vec[kk][q] * vec[kk][q]);
        scl[kk] = (float)(1. / (sqrt(square_mag) + DBL_EPSILON));
        const vfsms_keypoint kp = R.kps[k];
        R.kps_out[pos] = kp;
        R.kps_xy[2 * pos] = kp.x; R.kps_xy[2 * pos + 1] = kp.y;
    }
    __syncthreads();
    if (pos >= 0) {
        const float sc = scl[kk];
        for (int q = c; q < dsize; q += 16) R.desc[(size_t)pos * dsize + q] = vec[kk][q] * sc;
    }
}

// ---------------------------------------------------------------------------------------------------
// host side: arena carving and launch sequences
// ---------------------------------------------------------------------------------------------------
static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

size_t surf_roi_bytes(int h, int w, int cap, int nlayers_total, int noctaves, int dim)
{
    size_t b = al(sizeof(int32_t) * (size_t)(h + 1) * (w + 1)) + al(integral_carry_bytes(h, w)) + al(sizeof(uint16_t) * (size_t)h * w + 8);
    int lpo = nlayers_total / noctaves;
    for (int o = 0; o < noctaves; o++) {
        size_t n = (size_t)(h >> o) * (w >> o);
        b += lpo * al(sizeof(float) * (n ? n : 1));
    }
    b += al(16 * sizeof(int)) + al(sizeof(Cand) * cap) + al(sizeof(vfsms_keypoint) * cap) + al((size_t)cap * VFSMS_PATCH_ROW);
    b += 2 * al(sizeof(int) * cap) + al(sizeof(float) * 2 * cap) + al(sizeof(float) * (size_t)cap * dim) + al(sizeof(vfsms_keypoint) * cap);
    b += 2 * al(sizeof(DescRec) * (size_t)cap);               // this ROI's share of the launch's two record arrays (launch_surf_describe)
    return b + 4096;
}

int surf_roi_carve(vfsms_ctx *ctx, RoiDev *r, const uint8_t *img, int stride, int h, int w, int cap,
                   const vfsms_surf_params *p)
{
    const int lpo = p->n_octave_layers + 2;
    memset(r, 0, sizeof(*r));
    r->img = img; r->stride = stride; r->h = h; r->w = w; r->cap = cap;
    r->sum = (int32_t *)ctx_arena_alloc(ctx, sizeof(int32_t) * (size_t)(h + 1) * (w + 1));
    r->ipitch = (w + 3) & ~3;
    r->icarry = (int32_t *)ctx_arena_alloc(ctx, integral_carry_bytes(h, w));
    r->pair = (uint16_t *)ctx_arena_alloc(ctx, sizeof(uint16_t) * (size_t)h * w + 8);
    int step = 1;
    for (int o = 0; o < p->n_octaves; o++) {
        size_t n = (size_t)(h / step) * (w / step);
        for (int l = 0; l < lpo; l++) {
            r->det[o * lpo + l] = (float *)ctx_arena_alloc(ctx, sizeof(float) * (n ? n : 1));
        }
        step *= 2;
    }
    r->counters = (int *)ctx_arena_alloc(ctx, 16 * sizeof(int));
    r->cand = (Cand *)ctx_arena_alloc(ctx, sizeof(Cand) * cap);
    r->kps = (vfsms_keypoint *)ctx_arena_alloc(ctx, sizeof(vfsms_keypoint) * cap);
    r->patch = (uint8_t *)ctx_arena_alloc(ctx, (size_t)cap * VFSMS_PATCH_ROW);
    r->keep_pos = (int *)ctx_arena_alloc(ctx, sizeof(int) * cap);
    r->order = (int *)ctx_arena_alloc(ctx, sizeof(int) * cap);
    r->kps_xy = (float *)ctx_arena_alloc(ctx, sizeof(float) * 2 * cap);
    r->desc = (float *)ctx_arena_alloc(ctx, sizeof(float) * (size_t)cap * (p->extended ? 128 : 64));
    r->kps_out = (vfsms_keypoint *)ctx_arena_alloc(ctx, sizeof(vfsms_keypoint) * cap);
    if (!r->kps_out) { vfsms_set_error("arena exhausted while carving a SURF ROI"); return VFSMS_ERR_CAPACITY; }
    return VFSMS_OK;
}

int launch_surf_detect(vfsms_ctx *ctx, const RoiDev *d_rois, const RoiDev *h_rois, int nrois,
                       const vfsms_surf_params *p)
{
    if (nrois <= 0) return VFSMS_OK;
    const int lpo = p->n_octave_layers + 2;
    int maxcap = 0;
    // (the caller zeroes the ROI counters; det layers need no clearing: the non-maximum search only ever reads cells
    //  that k_hessian wrote -- its margins are those of the layer above -- so the 53 B/px layer memset is gone)
    for (int r = 0; r < nrois; r++) maxcap = h_rois[r].cap > maxcap ? h_rois[r].cap : maxcap;
    const std::vector<ShapeRun> runs = shape_runs(h_rois, nrois);
    {
        ProfScope ps(ctx, "integral");
        for (const ShapeRun &q : runs) TRY(launch_integral(ctx, d_rois + q.first, q.count, q.h, q.w));
    }
    {
        ProfScope ps(ctx, "hessian");
        static const bool generic = getenv("VFSMS_HESSIAN_GENERIC") && atoi(getenv("VFSMS_HESSIAN_GENERIC")) != 0;
        for (const ShapeRun &q : runs) {
            const RoiDev *dq = d_rois + q.first;
            int step = 1, o = 0;
            if (lpo == 5) {                                                // the two fine octaves: LDS-tiled
                for (; o < p->n_octaves && o < 2; o++) {
                    const int lrows = q.h / step, lcols = q.w / step;
                    if (lrows > 0 && lcols > 0) {
                        if (o == 0)
                            hipLaunchKernelGGL((k_hessian_lds<1, 64, 4>), dim3((lcols + 63) / 64, (lrows + 15) / 16, q.count), dim3(64, 4), 0, ctx->stream, dq, ctx->d_layers, lpo, o);
                        else
                            hipLaunchKernelGGL((k_hessian_lds<2, 32, HESS_O1_WAVES>), dim3((lcols + 31) / 32, (lrows + 15) / 16, q.count), dim3(64, HESS_O1_WAVES), 0, ctx->stream, dq, ctx->d_layers, lpo, o);
                    }
                    step *= 2;
                }
            }
            static const bool rows2 = !(getenv("VFSMS_HESSIAN_ROWS2") && atoi(getenv("VFSMS_HESSIAN_ROWS2")) == 0);
            if (lpo == 5 && o == 2 && o < p->n_octaves && rows2 && !generic) {      // octave 2: row-staged (k_hessian_rows2)
                const int s0 = 36;                                           // the octave's smallest layer has the most samples
                if (q.h >= s0 && q.w >= s0) {
                    const int tiles_i = ((q.h - s0) / 4 + 1 + 1) / 2, tiles_j = ((q.w - s0) / 4 + 1 + 127) / 128;
                    hipLaunchKernelGGL(k_hessian_rows2, dim3((unsigned)(tiles_i * tiles_j * 5 * q.count)), dim3(64, 4), 0, ctx->stream, dq, ctx->d_layers, lpo,
                                       tiles_i, tiles_j, q.count);
                }
                o++; step *= 2;
            }
            HessPlan plan; plan.o0 = o; plan.noct = 0; plan.first[0] = 0;
            for (; o < p->n_octaves && plan.noct < VFSMS_MAX_OCTAVES; o++) {
                const int lrows = q.h / step, lcols = q.w / step;
                const int tx = (lcols + 63) / 64, ty = (lrows + 3) / 4;
                plan.tiles_x[plan.noct] = tx > 0 ? tx : 1;
                plan.first[plan.noct + 1] = plan.first[plan.noct] + tx * ty;
                plan.noct++;
                step *= 2;
            }
            if (plan.noct > 0 && plan.first[plan.noct] > 0) {
                if (lpo == 5 && plan.o0 >= 2 && plan.o0 + plan.noct <= 4 && !generic)       // the stock pyramid: constant-offset taps
                    hipLaunchKernelGGL(k_hessian_coarse, dim3((unsigned)plan.first[plan.noct] * lpo * q.count), dim3(64, 4), 0, ctx->stream, dq, ctx->d_layers, lpo, plan, q.count);
                else
                    hipLaunchKernelGGL(k_hessian, dim3((unsigned)plan.first[plan.noct] * lpo * q.count), dim3(64, 4), 0, ctx->stream, dq, ctx->d_layers, lpo, plan, q.count);
            }
        }
    }
    {
        ProfScope ps(ctx, "nms");
        for (const ShapeRun &q : runs) {
            NmsPlan plan; plan.noct = 0; plan.first[0] = 0;
            int step = 1;
            for (int o = 0; o < p->n_octaves && o < VFSMS_MAX_OCTAVES; o++) {
                const int lrows = q.h / step, lcols = q.w / step;
                const int tx = (lcols + 63) / 64, ty = (lrows + NMS_TH - 1) / NMS_TH;
                plan.tiles_x[o] = tx > 0 ? tx : 1;
                plan.first[o + 1] = plan.first[o] + tx * ty;
                plan.noct = o + 1;
                step *= 2;
            }
            if (plan.noct > 0 && plan.first[plan.noct] > 0)
                hipLaunchKernelGGL(k_nms, dim3(plan.first[plan.noct], q.count * p->n_octave_layers), dim3(64, 4), 0, ctx->stream, d_rois + q.first,
                                   ctx->d_layers, lpo, p->n_octave_layers, plan, p->hessian_threshold);
        }
    }
    {
        ProfScope ps(ctx, "sort");
        static const bool n2sort = getenv("VFSMS_SORT_N2") && atoi(getenv("VFSMS_SORT_N2")) != 0;
        int mincap = maxcap;
        for (int r = 0; r < nrois; r++) mincap = h_rois[r].cap < mincap ? h_rois[r].cap : mincap;
        if (n2sort || mincap < SORT_BUCKETS) hipLaunchKernelGGL(k_rank_sort, dim3((maxcap + 63) / 64, nrois), dim3(256), 0, ctx->stream, d_rois, ctx->d_layers);
        else {
            hipLaunchKernelGGL(k_bucket_sort, dim3(nrois), dim3(1024), 0, ctx->stream, d_rois);
            hipLaunchKernelGGL(k_bucket_rank, dim3((maxcap + 255) / 256, nrois), dim3(256), 0, ctx->stream, d_rois, ctx->d_layers);
        }
    }
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}


// ---- INTER_AREA tables (computeResizeAreaTab for every descriptor-window size, once per context) -----------------------------------
int ctx_prepare_area_tab(vfsms_ctx *ctx)
{
    if (ctx->d_area_tab) return VFSMS_OK;
    std::vector<AreaRec> tab((size_t)(VFSMS_MAX_WIN + 1) * AREA_RECS);
    memset(tab.data(), 0, sizeof(AreaRec) * tab.size());
    const int dsz = 21;
    for (int win = dsz; win <= VFSMS_MAX_WIN; win++) {
        AreaRec *rec = tab.data() + (size_t)win * AREA_RECS;
        const double inv_scale = (double)dsz / win;
        const double scale = 1. / inv_scale;
        const int iscale = (int)rint(scale);
        const bool fast = fabs(scale - iscale) < DBL_EPSILON;
        int nmin = 1 << 30, nmax = 0;
        for (int dx = 0; dx < dsz; dx++) {
            AreaRec &c = rec[dx];
            if (fast) { c.j0 = dx * iscale; c.n = iscale; c.a0 = c.af = c.al = 1.f; }
            else {
                // one destination index of computeResizeAreaTab: left partial, full cells [sx1, sx2), right partial
                const double fsx1 = dx * scale;
                const double fsx2 = fsx1 + scale;
                const double cellWidth = fmin(scale, win - fsx1);
                int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
                sx2 = std::min(sx2, win - 1);
                sx1 = std::min(sx1, sx2);
                const bool left = sx1 - fsx1 > 1e-3, right = fsx2 - sx2 > 1e-3;
                const float a_left = (float)((sx1 - fsx1) / cellWidth), a_full = (float)(1.0 / cellWidth);
                const float a_right = (float)(fmin(fmin(fsx2 - sx2, 1.), cellWidth) / cellWidth);
                c.j0 = left ? sx1 - 1 : sx1;
                c.n = (left ? 1 : 0) + (sx2 - sx1) + (right ? 1 : 0);
                c.a0 = left ? a_left : (sx2 > sx1 ? a_full : a_right);
                c.af = a_full;
                c.al = right ? a_right : a_full;
                if (c.n == 1 && left) c.al = a_left;
            }
            if (c.n < 1 || c.j0 < 0 || c.j0 + c.n > win) { vfsms_set_error("internal: INTER_AREA table of window %d", win); return VFSMS_ERR_UNSUPPORTED; }
            nmin = std::min(nmin, c.n); nmax = std::max(nmax, c.n);
        }
        if (nmax > 40) { vfsms_set_error("internal: INTER_AREA run of window %d", win); return VFSMS_ERR_UNSUPPORTED; }
        rec[21].j0 = nmin; rec[21].n = nmax; rec[21].a0 = fast ? 1.f / (iscale * iscale) : 1.f;
        rec[21].mode = !fast ? 0 : iscale == 2 ? 2 : 1;
    }
    for (int win = 0; win < dsz; win++) memcpy(tab.data() + (size_t)win * AREA_RECS, tab.data() + (size_t)dsz * AREA_RECS, sizeof(AreaRec) * AREA_RECS);
    HIP_TRY(hipMalloc((void **)&ctx->d_area_tab, sizeof(AreaRec) * tab.size()));
    HIP_TRY(hipMemcpy(ctx->d_area_tab, tab.data(), sizeof(AreaRec) * tab.size(), hipMemcpyHostToDevice));
    return VFSMS_OK;
}

int launch_surf_describe(vfsms_ctx *ctx, const RoiDev *d_rois, const RoiDev *h_rois, int nrois,
                         const vfsms_surf_params *p)
{
    if (nrois <= 0) return VFSMS_OK;
    if (nrois > VFSMS_MAX_ROIS) { vfsms_set_error("more than %d ROIs in one batch", VFSMS_MAX_ROIS); return VFSMS_ERR_CAPACITY; }
    TRY(ctx_prepare_area_tab(ctx));
    int maxcap = 0;
    for (int r = 0; r < nrois; r++) maxcap = h_rois[r].cap > maxcap ? h_rois[r].cap : maxcap;
    {
        ProfScope ps(ctx, "orientation");
        hipLaunchKernelGGL(k_orientation, dim3((maxcap + ORI_KP - 1) / ORI_KP, nrois), dim3(ORI_KP * 128), 0, ctx->stream, d_rois, ctx->d_tables, p->upright);
    }
    {
        ProfScope ps(ctx, "compact");
        hipLaunchKernelGGL(k_keep_scan, dim3(nrois), dim3(1024), 0, ctx->stream, d_rois);
    }
    {
        ProfScope ps(ctx, "describe");
        // ticket heads (8 counters per kernel, 256 B apart), the launch's plan and its two record arrays out of the call's arena (surf_roi_bytes
        // counts the records of every ROI; callers reserve 64 KB of slack for the rest)
        int *tickets = (int *)ctx_arena_alloc(ctx, sizeof(int) * 2 * DESC_HEADS * DESC_HEAD_STRIDE);
        DescPlan *plan = (DescPlan *)ctx_arena_alloc(ctx, sizeof(DescPlan));
        size_t capsum = 0;
        for (int r = 0; r < nrois; r++) capsum += (size_t)h_rois[r].cap;
        DescRec *rec_big = (DescRec *)ctx_arena_alloc(ctx, sizeof(DescRec) * capsum);
        DescRec *rec_small = (DescRec *)ctx_arena_alloc(ctx, sizeof(DescRec) * capsum);
        if (!tickets || !plan || !rec_big || !rec_small) { vfsms_set_error("arena exhausted (descriptor work list)"); return VFSMS_ERR_CAPACITY; }
        HIP_TRY(hipMemsetAsync(tickets, 0, sizeof(int) * 2 * DESC_HEADS * DESC_HEAD_STRIDE, ctx->stream));
        for (const ShapeRun &q : shape_runs(h_rois, nrois))
            hipLaunchKernelGGL(k_pair_rows, dim3((q.w + 1023) / 1024, q.h, q.count), dim3(256), 0, ctx->stream, d_rois + q.first);
        hipLaunchKernelGGL(k_desc_order, dim3(nrois), dim3(1024), 0, ctx->stream, d_rois);
        hipLaunchKernelGGL(k_desc_plan, dim3(1), dim3(256), 0, ctx->stream, d_rois, nrois, plan, 256 * DESC_WGS);
        hipLaunchKernelGGL(k_desc_recs, dim3((maxcap + 255) / 256, nrois), dim3(256), 0, ctx->stream, d_rois, plan, rec_big, rec_small, p->upright);
        hipLaunchKernelGGL(k_describe, dim3(256 * DESC_WGS), dim3(256), 0, ctx->stream, d_rois, plan, rec_big, tickets,
                           ctx->d_tables, (const AreaRec *)ctx->d_area_tab, p->extended, p->upright);
        hipLaunchKernelGGL(k_describe_small, dim3(256 * DESC_SMALL_WGS), dim3(256), 0, ctx->stream, d_rois, plan, rec_small,
                           tickets + DESC_HEADS * DESC_HEAD_STRIDE, (const AreaRec *)ctx->d_area_tab, p->upright);
        hipLaunchKernelGGL(k_desc_tail, dim3((maxcap + 15) / 16, nrois), dim3(256), 0, ctx->stream, d_rois, ctx->d_tables, p->extended);
    }
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

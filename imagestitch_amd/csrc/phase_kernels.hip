// phase_kernels.hip -- FP64 phase correlation for gfx950: native rocFFT real-forward / real-inverse plans, BATCHED over the
// attempts of a launch (one plan execution transforms every ROI of a batch), with hand-written batched pad / cross-power /
// shifted-argmax / centroid kernels around them; since round 6 the transforms themselves run in LDS for the strip shapes of this path.
//
// Replaces cv2.phaseCorrelate(np.float64(roiA), np.float64(roiB)) at Stitcher.py:230, i.e. OpenCV 3.3.1
// imgproc/src/phasecorr.cpp semantics: zero-pad bottom/right to
// getOptimalDFTSize, P = F1 * conj(F2), C = idft(P / |P|) unscaled with the packed-format quirk that the
// purely-real bins (DC / Nyquist) divide by x*x instead of |x|, fftShift by quadrant swap (odd sizes keep
// their last row/column), first-maximum argmax, 5x5 clamped weighted centroid, response / (M*N).
// All of it is HBM-bound streaming work; the u8 -> f64 conversion is fused into the pad kernel and the
// quadrant swap is never materialised (the argmax and centroid kernels index through it).
//
// Batch layout for nb attempts of one ROI size (padded M x N, Nc = N/2 + 1):
//   RE : 2 nb real planes  [a0, b0, a1, b1, ...]  -> forward plan with 2 nb transforms -> FQ : 2 nb half spectra
//   CP : nb cross-power spectra (compact)         -> inverse plan with nb transforms   -> RE (first nb planes reused)
// Plans are cached per (M, N, batch) for batches of up to 32 attempts; larger batches run in chunks of 32.
#include "common.h"
#include <rocfft/rocfft.h>
#include <float.h>
#include <string.h>
#include <algorithm>
#include <stdlib.h>
#include <math.h>
#include <vector>

#define PHASE_MAX_CHUNK 32

static int optimal_dft_size(int n)
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

struct PhaseJob { const uint8_t *a, *b; int sa, sb; };

// u8 ROI pair -> two zero-padded FP64 planes; blockIdx.z = job
__global__ __launch_bounds__(256) void k_pad_u8_f64(const PhaseJob *__restrict__ jobs, int h, int w, int M, int N, double *__restrict__ RE)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= N) return;
    const PhaseJob J = jobs[blockIdx.z];
    const bool in = (y < h) && (x < w);
    const size_t plane = (size_t)M * N;
    const size_t o = (size_t)(2 * blockIdx.z) * plane + (size_t)y * N + x;
    RE[o] = in ? (double)J.a[(size_t)y * J.sa + x] : 0.0;
    RE[o + plane] = in ? (double)J.b[(size_t)y * J.sb + x] : 0.0;
}

// mulSpectrums(conjB) + magSpectrums + divSpectrums on the half spectra (M x (N/2+1) complex each); blockIdx.z = job
struct cplx { double x, y; };
__global__ __launch_bounds__(256) void k_cross_power(const cplx *__restrict__ FQ, cplx *__restrict__ CP, int M, int N)
{
    const int Nc = N / 2 + 1;
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int u = blockIdx.y;
    if (v >= Nc) return;
    const size_t plane = (size_t)M * Nc;
    const size_t k = (size_t)u * Nc + v;
    const cplx f1 = FQ[(size_t)(2 * blockIdx.z) * plane + k], f2 = FQ[(size_t)(2 * blockIdx.z + 1) * plane + k];
    const double ar = f1.x, ai = f1.y, br = f2.x, bi = f2.y;
    const double pr = ar * br + ai * bi;
    const double pi = ai * br - ar * bi;
    const bool real_u = (u == 0) || ((M % 2 == 0) && u == M / 2);
    const bool real_v = (v == 0) || ((N % 2 == 0) && v == N / 2);
    const double eps = DBL_EPSILON;
    cplx c;
    if (real_u && real_v) {
        const double mg = pr * pr;
        c.x = pr / (mg + eps); c.y = 0.0;
    } else {
        const double mg = sqrt(pr * pr + pi * pi);
        const double denom = mg * mg + eps;
        c.x = (pr * mg) / denom;
        c.y = (pi * mg) / denom;
    }
    CP[(size_t)blockIdx.z * plane + k] = c;
}

// shifted (ys, xs) -> source (y, x) under phasecorr.cpp's fftShift (OpenCV 3.3.1): the four quadrants of (M >> 1) x (N >> 1) swap diagonally; an
// odd LAST row or column is outside every quadrant and stays where it is -- entirely: its elements move along NEITHER axis (until round 6 the
// two axes were unshifted one by one, which moved the elements of an odd last row along x: wrong whenever the peak or its 5 x 5 window touched
// that row or column; tools/phase_ab.py's 5 x 7 strip found it).  A 1 x n or n x 1 surface swaps its two halves.  The map is its own inverse.
__device__ __forceinline__ void unshift2(int ys, int xs, int M, int N, int &y, int &x)
{
    const int ym = M >> 1, xm = N >> 1;
    y = ys; x = xs;
    if (ym == 0 || xm == 0) {
        const int mid = (M * N) >> 1;
        int i = ys * N + xs;
        i = i < mid ? i + mid : (i < 2 * mid ? i - mid : i);
        y = i / N; x = i - y * N;
    } else if (ys < 2 * ym && xs < 2 * xm) {
        y = ys < ym ? ys + ym : ys - ym;
        x = xs < xm ? xs + xm : xs - xm;
    }
}

struct ArgMax { double v; long long idx; };

__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b)
{
    // larger value wins; equal values -> smaller shifted index (minMaxLoc returns the first maximum)
    if (b.v > a.v || (b.v == a.v && b.idx < a.idx)) return b;
    return a;
}

#define PHASE_NBLK 64       // partial argmax blocks per job (each walks whole rows: no per-element division)

__global__ __launch_bounds__(256) void k_argmax_partial(const double *__restrict__ RE, int M, int N, ArgMax *partial)
{
    const double *R = RE + (size_t)blockIdx.y * M * N;
    ArgMax best; best.v = -INFINITY; best.idx = (long long)M * N;
    for (int ys = blockIdx.x; ys < M; ys += PHASE_NBLK) {
        for (int xs = threadIdx.x; xs < N; xs += 256) {
            int y, x;
            unshift2(ys, xs, M, N, y, x);
            ArgMax c; c.v = R[(size_t)y * N + x]; c.idx = (long long)ys * N + xs;
            best = better(best, c);
        }
    }
    __shared__ ArgMax sm[256];
    sm[threadIdx.x] = best;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) sm[threadIdx.x] = better(sm[threadIdx.x], sm[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * PHASE_NBLK + blockIdx.x] = sm[0];
}

__global__ __launch_bounds__(64) void k_argmax_centroid(const double *__restrict__ RE, int M, int N, const ArgMax *partial, double *out3)
{
    const double *R = RE + (size_t)blockIdx.x * M * N;
    __shared__ ArgMax sm[64];
    sm[threadIdx.x] = partial[(size_t)blockIdx.x * PHASE_NBLK + threadIdx.x];
    __syncthreads();
    for (int d = 32; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) sm[threadIdx.x] = better(sm[threadIdx.x], sm[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const int py = (int)(sm[0].idx / N), px = (int)(sm[0].idx % N);
    int minr = py - 2, maxr = py + 2, minc = px - 2, maxc = px + 2;     // weightedCentroid, 5x5, clamped
    if (minr < 0) minr = 0;
    if (minc < 0) minc = 0;
    if (maxr > M - 1) maxr = M - 1;
    if (maxc > N - 1) maxc = N - 1;
    double cx = 0, cy = 0, s = 0;
    for (int y = minr; y <= maxr; y++)
        for (int x = minc; x <= maxc; x++) {
            int uy, ux;
            unshift2(y, x, M, N, uy, ux);
            const double v = R[(size_t)uy * N + ux];
            cx += (double)x * v; cy += (double)y * v; s += v;
        }
    double response = s;
    s += DBL_EPSILON;
    cx /= s; cy /= s;
    response /= (double)M * (double)N;
    double *o = out3 + 3 * (size_t)blockIdx.x;
    o[0] = (double)N / 2.0 - cx;
    o[1] = (double)M / 2.0 - cy;
    o[2] = response;
}

// ---- the transforms in LDS (round 6) --------------------------------------------------------------------------------------------------------
// rocFFT runs a 2-D real plan as row transforms + a transpose + column transforms + a transpose (and the same again for the inverse): 39 % of the
// phase path's time went into three transpose kernels (profiles/r06_kernel_stats_phase.csv), every one a full pass over the FP64 spectra in HBM.
// The padded strips of this path are short in one direction (0.2 x the tile: 432 of 2048, 864 of 4096), so a column of the spectrum FITS LDS
// several times over and no transpose is needed at all:
//   k_phase_rows_fwd   u8 row -> N/2-point complex Stockham transform in LDS (the even/odd packing of a real row) -> the N/2+1 bins of the row;
//                      rows below the strip (zero padding) are neither transformed nor written;
//   k_phase_cols       C adjacent columns of BOTH spectra -> forward column transforms -> mulSpectrums(conj) / magnitude / divide in LDS
//                      (k_cross_power's arithmetic) -> inverse column transforms -> the cross-power columns; the two forward spectra are
//                      read once and never written back;
//   k_phase_rows_inv   N/2+1 bins of a row -> the packed N/2-point inverse -> N reals + the row block's shifted arg-max;
//   k_peak_centroid    the partial arg-max records of a job -> first maximum -> 5 x 5 centroid.
// A strip that is TALL (a left-right neighbour: 2048 x 409) is transposed first as bytes (0.8 MB) and correlated as its transpose -- the
// correlation surface of the transposes is the transpose of the surface, cross power and peak pick are symmetric in the two axes, and the
// arg-max / centroid index the surface through the original coordinates -- so that the long axis is always the contiguous one.
// Lengths: every 2^a 3^b 5^c that fits (radix 4 / 2 / 3 / 5 passes, one LDS buffer: all inputs of a pass are in registers before a barrier,
// all outputs written after it); an odd row length or a column too long for LDS takes the rocFFT path below.
#ifndef PHASE_TW_LOAD
#define PHASE_TW_LOAD 0  // 1: the R - 1 twiddles of a radix >= 8 butterfly are gathered from the (L1 / L2 resident) table instead of formed by squaring from one
#endif
struct FftPass { int R, Ls, m, twstride; unsigned magic_m, magic_Ls; };      // m = L / R butterflies per transform; Ls = length done so far
struct FftSched { int n, L; unsigned magic_L, magic_L1; FftPass p[13]; };      // magic_L / magic_L1: x / L and x / (L + 1) by __umulhi

__device__ __forceinline__ cplx c_add(cplx a, cplx b) { cplx r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
__device__ __forceinline__ cplx c_sub(cplx a, cplx b) { cplx r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }
__device__ __forceinline__ cplx c_mul(cplx a, cplx w) { cplx r; r.x = fma(a.x, w.x, -(a.y * w.y)); r.y = fma(a.x, w.y, a.y * w.x); return r; }
__device__ __forceinline__ cplx c_muli(cplx a, double s) { cplx r; r.x = -s * a.y; r.y = s * a.x; return r; }      // s * i * a, s = +-1

template <int R> __device__ __forceinline__ void bfly(cplx *v, double s);      // DFT of R values in place, kernel exp(s 2 pi i / R)
template <> __device__ __forceinline__ void bfly<2>(cplx *v, double)
{
    const cplx a = v[0], b = v[1];
    v[0] = c_add(a, b); v[1] = c_sub(a, b);
}
template <> __device__ __forceinline__ void bfly<3>(cplx *v, double s)
{
    const cplx a = v[0], t1 = c_add(v[1], v[2]), d = c_sub(v[1], v[2]);
    cplx t2; t2.x = fma(-0.5, t1.x, a.x); t2.y = fma(-0.5, t1.y, a.y);
    cplx t3; t3.x = d.x * 0.86602540378443864676; t3.y = d.y * 0.86602540378443864676;
    const cplx it = c_muli(t3, s);
    v[0] = c_add(a, t1); v[1] = c_add(t2, it); v[2] = c_sub(t2, it);
}
template <> __device__ __forceinline__ void bfly<4>(cplx *v, double s)
{
    const cplx p = c_add(v[0], v[2]), q = c_sub(v[0], v[2]), r = c_add(v[1], v[3]), u = c_muli(c_sub(v[1], v[3]), s);
    v[0] = c_add(p, r); v[1] = c_add(q, u); v[2] = c_sub(p, r); v[3] = c_sub(q, u);
}
template <> __device__ __forceinline__ void bfly<5>(cplx *v, double s)
{
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410, s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
    const cplx a = v[0], t1 = c_add(v[1], v[4]), t2 = c_add(v[2], v[3]), t3 = c_sub(v[1], v[4]), t4 = c_sub(v[2], v[3]);
    cplx m1, m2, n1, n2;
    m1.x = fma(c2, t2.x, fma(c1, t1.x, a.x)); m1.y = fma(c2, t2.y, fma(c1, t1.y, a.y));
    m2.x = fma(c1, t2.x, fma(c2, t1.x, a.x)); m2.y = fma(c1, t2.y, fma(c2, t1.y, a.y));
    n1.x = fma(s2, t4.x, s1 * t3.x); n1.y = fma(s2, t4.y, s1 * t3.y);
    n2.x = fma(-s1, t4.x, s2 * t3.x); n2.y = fma(-s1, t4.y, s2 * t3.y);
    const cplx i1 = c_muli(n1, s), i2 = c_muli(n2, s);
    v[0] = c_add(a, c_add(t1, t2));
    v[1] = c_add(m1, i1); v[4] = c_sub(m1, i1); v[2] = c_add(m2, i2); v[3] = c_sub(m2, i2);
}

// exp(s 2 pi i e / N) for the in-register composite butterflies (N = 6, 8, 9, 10, 16): correctly rounded literals; e is a compile-time constant
// after unrolling, so the lookups fold into immediates
__device__ __forceinline__ cplx unit_root(int N, int e, double s)
{
    constexpr double C6[6] = {1.0, 0.5, -0.5, -1.0, -0.5, 0.5};
    constexpr double S6[6] = {0.0, 0.8660254037844386, 0.8660254037844386, 0.0, -0.8660254037844386, -0.8660254037844386};
    constexpr double C8[8] = {1.0, 0.7071067811865476, 0.0, -0.7071067811865476, -1.0, -0.7071067811865476, 0.0, 0.7071067811865476};
    constexpr double S8[8] = {0.0, 0.7071067811865476, 1.0, 0.7071067811865476, 0.0, -0.7071067811865476, -1.0, -0.7071067811865476};
    constexpr double C9[9] = {1.0, 0.766044443118978, 0.17364817766693036, -0.5, -0.9396926207859084, -0.9396926207859084, -0.5, 0.17364817766693036, 0.766044443118978};
    constexpr double S9[9] = {0.0, 0.6427876096865394, 0.984807753012208, 0.8660254037844386, 0.3420201433256687, -0.3420201433256687, -0.8660254037844386,
                              -0.984807753012208, -0.6427876096865394};
    constexpr double C10[10] = {1.0, 0.8090169943749475, 0.30901699437494745, -0.30901699437494745, -0.8090169943749475, -1.0, -0.8090169943749475,
                                -0.30901699437494745, 0.30901699437494745, 0.8090169943749475};
    constexpr double S10[10] = {0.0, 0.5877852522924731, 0.9510565162951535, 0.9510565162951535, 0.5877852522924731, 0.0, -0.5877852522924731,
                                -0.9510565162951535, -0.9510565162951535, -0.5877852522924731};
    constexpr double C16[16] = {1.0, 0.9238795325112867, 0.7071067811865476, 0.3826834323650898, 0.0, -0.3826834323650898, -0.7071067811865476, -0.9238795325112867,
                                -1.0, -0.9238795325112867, -0.7071067811865476, -0.3826834323650898, 0.0, 0.3826834323650898, 0.7071067811865476, 0.9238795325112867};
    constexpr double S16[16] = {0.0, 0.3826834323650898, 0.7071067811865476, 0.9238795325112867, 1.0, 0.9238795325112867, 0.7071067811865476, 0.3826834323650898,
                                0.0, -0.3826834323650898, -0.7071067811865476, -0.9238795325112867, -1.0, -0.9238795325112867, -0.7071067811865476, -0.3826834323650898};
    e %= N;
    cplx w;
    if (N == 6) { w.x = C6[e]; w.y = s * S6[e]; }
    else if (N == 8) { w.x = C8[e]; w.y = s * S8[e]; }
    else if (N == 9) { w.x = C9[e]; w.y = s * S9[e]; }
    else if (N == 10) { w.x = C10[e]; w.y = s * S10[e]; }
    else { w.x = C16[e]; w.y = s * S16[e]; }
    return w;
}

// DFT of R1 * R2 values in registers: R2 transforms of length R1 over v[a + R2 b], the twiddles W^(a k1), R1 transforms of length R2 -> v[k1 + R1 k2]
template <int R1, int R2> __device__ __forceinline__ void bfly_ct(cplx *v, double s)
{
    cplx y[R2][R1];
#pragma unroll
    for (int a = 0; a < R2; a++) {
#pragma unroll
        for (int b = 0; b < R1; b++) y[a][b] = v[a + R2 * b];
        bfly<R1>(y[a], s);
        if (a > 0) {
#pragma unroll
            for (int k1 = 1; k1 < R1; k1++) y[a][k1] = c_mul(y[a][k1], unit_root(R1 * R2, a * k1, s));
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < R1; k1++) {
        cplx z[R2];
#pragma unroll
        for (int a = 0; a < R2; a++) z[a] = y[a][k1];
        bfly<R2>(z, s);
#pragma unroll
        for (int k2 = 0; k2 < R2; k2++) v[k1 + R1 * k2] = z[k2];
    }
}
template <> __device__ __forceinline__ void bfly<6>(cplx *v, double s) { bfly_ct<3, 2>(v, s); }
template <> __device__ __forceinline__ void bfly<8>(cplx *v, double s) { bfly_ct<4, 2>(v, s); }
template <> __device__ __forceinline__ void bfly<9>(cplx *v, double s) { bfly_ct<3, 3>(v, s); }
template <> __device__ __forceinline__ void bfly<10>(cplx *v, double s) { bfly_ct<5, 2>(v, s); }
template <> __device__ __forceinline__ void bfly<16>(cplx *v, double s) { bfly_ct<4, 4>(v, s); }

// v[t] *= w^t, t = 1 .. R-1, from the one table entry w: powers by squaring (at most three products deep), so a butterfly costs ONE twiddle
// gather from LDS instead of R - 1
template <int R> __device__ __forceinline__ void apply_powers(cplx *v, cplx w1)
{
    if (R <= 4) {
        cplx w = w1;
        v[1] = c_mul(v[1], w);
#pragma unroll
        for (int t = 2; t < R; t++) { w = c_mul(w, w1); v[t] = c_mul(v[t], w); }
    } else {
        const cplx w2 = c_mul(w1, w1), w4 = c_mul(w2, w2), w8 = c_mul(w4, w4);
#pragma unroll
        for (int t = 1; t < R; t++) {
            cplx w = (t & 1) ? w1 : ((t & 2) ? w2 : ((t & 4) ? w4 : w8));
            const int low = (t & 1) ? 1 : ((t & 2) ? 2 : ((t & 4) ? 4 : 8));
            if ((t & 2) && low < 2) w = c_mul(w, w2);
            if ((t & 4) && low < 4) w = c_mul(w, w4);
            if ((t & 8) && low < 8) w = c_mul(w, w8);
            v[t] = c_mul(v[t], w);
        }
    }
}

// LDS layout of a transform: element e at e + (e >> 4).  The first pass (Ls = 1) writes the R outputs of butterfly j to j R .. j R + R - 1: at R = 16
// the lanes of a wave would all fall on the same four banks (a 64-way conflict on each of the sixteen ds_write_b128); one point of padding per 16
// spreads them over all banks.  Transforms lie fft_lp(L) apart.
__device__ __forceinline__ int pad_e(int e) { return e + (e >> 4); }
__host__ __device__ __forceinline__ int fft_lp(int L) { return L + (L >> 4) + 1; }

// One Stockham pass over nfft transforms of length L in `buf` (LDS).  tw: global table exp(-2 pi i q / Ltab) with P.twstride = Ltab / (Ls R): ONE gather
// per butterfly (L1 / L2 resident, issued beside the LDS reads).  A thread holds at most NB butterflies of radix R (R NB >= 8): the host sizes the
// workgroup so that nfft * L <= 8 T.
template <int R, int NB>
__device__ __forceinline__ void fft_pass(cplx *buf, const cplx *__restrict__ tw, const FftPass P, int LP, int nfft, double s, int tid, int T)
{
    const int total = P.m * nfft;
    cplx v[NB][R];
    int ob[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const int id = tid + i * T;
        ob[i] = -1;
        if (id < total) {
            const int f = P.magic_m ? (int)__umulhi((unsigned)id, P.magic_m) : id;       // id / m (magic 0: m == 1)
            const int j = id - f * P.m;
            const int k = P.magic_Ls ? j - (int)__umulhi((unsigned)j, P.magic_Ls) * P.Ls : 0;   // j % Ls (magic 0: Ls == 1)
            const int kk = k * P.twstride;
            cplx w1; w1.x = 1.0; w1.y = 0.0;
            cplx wt[PHASE_TW_LOAD && R >= 8 ? R : 1];
            if (P.Ls > 1) {
                if (PHASE_TW_LOAD && R >= 8) {
#pragma unroll
                    for (int t = 1; t < R; t++) wt[PHASE_TW_LOAD && R >= 8 ? t : 0] = tw[t * kk];
                } else {
                    w1 = tw[kk];
                }
            }
            const cplx *src = buf + f * LP;
            int e = j;
#pragma unroll
            for (int t = 0; t < R; t++) { v[i][t] = src[pad_e(e)]; e += P.m; }
            if (P.Ls > 1) {
                if (PHASE_TW_LOAD && R >= 8) {
#pragma unroll
                    for (int t = 1; t < R; t++) {
                        cplx w = wt[PHASE_TW_LOAD && R >= 8 ? t : 0];
                        w.y = s > 0.0 ? -w.y : w.y;
                        v[i][t] = c_mul(v[i][t], w);
                    }
                } else {
                    w1.y = s > 0.0 ? -w1.y : w1.y;
                    apply_powers<R>(v[i], w1);
                }
            }
            bfly<R>(v[i], s);
            ob[i] = (j - k) * R + k + (f << 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NB; i++)
        if (ob[i] >= 0) {
            cplx *dst = buf + (ob[i] >> 16) * LP;
            int e = ob[i] & 0xffff;
#pragma unroll
            for (int t = 0; t < R; t++) { dst[pad_e(e)] = v[i][t]; e += P.Ls; }
        }
    __syncthreads();
}

__device__ __forceinline__ void fft_lds(cplx *buf, const cplx *tw, const FftSched &S, int LP, int nfft, double s, int tid, int T)
{
    for (int p = 0; p < S.n; p++) {
        const FftPass P = S.p[p];
        switch (P.R) {
        case 16: fft_pass<16, 1>(buf, tw, P, LP, nfft, s, tid, T); break;
        case 9: fft_pass<9, 1>(buf, tw, P, LP, nfft, s, tid, T); break;
        case 8: fft_pass<8, 1>(buf, tw, P, LP, nfft, s, tid, T); break;
        case 6: fft_pass<6, 2>(buf, tw, P, LP, nfft, s, tid, T); break;
        case 10: fft_pass<10, 1>(buf, tw, P, LP, nfft, s, tid, T); break;
        case 4: fft_pass<4, 2>(buf, tw, P, LP, nfft, s, tid, T); break;
        case 2: fft_pass<2, 4>(buf, tw, P, LP, nfft, s, tid, T); break;
        case 3: fft_pass<3, 3>(buf, tw, P, LP, nfft, s, tid, T); break;
        default: fft_pass<5, 2>(buf, tw, P, LP, nfft, s, tid, T); break;
        }
    }
}

// bytes of an h x w strip -> its transpose (w x h, row length h); blockIdx.z = 2 * job + image
__global__ __launch_bounds__(256) void k_phase_transpose_u8(const PhaseJob *__restrict__ jobs, int h, int w, uint8_t *__restrict__ out)
{
    __shared__ uint8_t tile[64][65];
    const PhaseJob J = jobs[blockIdx.z >> 1];
    const uint8_t *src = (blockIdx.z & 1) ? J.b : J.a;
    const int st = (blockIdx.z & 1) ? J.sb : J.sa;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int r = ly; r < 64; r += 4)
        tile[r][lx] = (y0 + r < h && x0 + lx < w) ? src[(size_t)(y0 + r) * st + x0 + lx] : (uint8_t)0;
    __syncthreads();
    uint8_t *dst = out + (size_t)blockIdx.z * h * w;
    for (int r = ly; r < 64; r += 4)
        if (x0 + r < w && y0 + lx < h) dst[(size_t)(x0 + r) * h + y0 + lx] = tile[lx][r];
}

__device__ __forceinline__ int div_magic(int x, unsigned magic) { return magic ? (int)__umulhi((unsigned)x, magic) : x; }   // x / d, x < 2^16 (magic 0: d == 1)

// The global loads of a workgroup's prologue are issued TOGETHER (fixed, unrolled trip counts; clamped addresses instead of branches): with the
// plain strided loops a thread waited out one HBM round trip per iteration, eight times in a row, with a handful of waves per CU to hide it.
#ifndef PHASE_DBG
#define PHASE_DBG 0      // timing experiments only (tools/job_r06_*.sh): 1 no forward column passes, 2 no inverse ones, 4 no cross power
#endif
#define PHASE_NI 8          // points of a workgroup per thread, at most (host: nfft * L <= 8 T)

// Forward row transforms: workgroup = RPW rows of one image (blockIdx.y = 2 * job + image), rows [0, hh) of length ww padded to N = 2 H.
// LDS: RPW transforms of H points (padded layout).  FQ: [2 nb][M][H + 1].
__global__ __launch_bounds__(512) void k_phase_rows_fwd(const PhaseJob *__restrict__ jobs, int hh, int ww, int M, int RPW, int lc, const FftSched S,
                                                         const cplx *__restrict__ tabN, cplx *__restrict__ FQ)
{
    extern __shared__ __align__(16) unsigned char phase_lds[];
    const int H = S.L, LP = fft_lp(H), T = blockDim.x, tid = threadIdx.x;
    cplx *buf = (cplx *)phase_lds;
    const PhaseJob J = jobs[blockIdx.y >> 1];
    const uint8_t *src = (blockIdx.y & 1) ? J.b : J.a;
    const int st = (blockIdx.y & 1) ? J.sb : J.sa;
    const int y0 = blockIdx.x * RPW;
    const int rows = min(RPW, hh - y0);
    const int npts = rows * H;
    {
        unsigned px0[PHASE_NI], px1[PHASE_NI];
#pragma unroll
        for (int i = 0; i < PHASE_NI; i++) {
            const int p = min(tid + i * T, npts - 1);
            const int r = div_magic(p, S.magic_L), n = p - r * H;
            const uint8_t *row = src + (size_t)(y0 + r) * st;
            px0[i] = row[min(2 * n, ww - 1)];
            px1[i] = row[min(2 * n + 1, ww - 1)];
        }
#pragma unroll
        for (int i = 0; i < PHASE_NI; i++) {
            const int p = tid + i * T;
            if (p < npts) {
                const int r = div_magic(p, S.magic_L), n = p - r * H;
                cplx z;
                z.x = (2 * n < ww) ? (double)px0[i] : 0.0;
                z.y = (2 * n + 1 < ww) ? (double)px1[i] : 0.0;
                buf[r * LP + pad_e(n)] = z;
            }
        }
    }
    __syncthreads();
    fft_lds(buf, tabN, S, LP, rows, -1.0, tid, T);
    // X[k] = ((Z[k] + conj Z[H-k]) - i e^(-2 pi i k / N) (Z[k] - conj Z[H-k])) / 2,  k = 0 .. H, Z[H] = Z[0]
    const int Nc = H + 1, nout = rows * Nc, C = 1 << lc;
    const size_t plane = (size_t)M * (((Nc + C - 1) >> lc) << lc);            // spectra are stored in column tiles of C: [tile][row][C]
    cplx wk[PHASE_NI + 1];
#pragma unroll
    for (int i = 0; i < PHASE_NI + 1; i++) {
        const int p = min(tid + i * T, nout - 1);
        const int r = div_magic(p, S.magic_L1);
        wk[i] = tabN[p - r * Nc];
    }
#pragma unroll
    for (int i = 0; i < PHASE_NI + 1; i++) {
        const int p = tid + i * T;
        if (p < nout) {
            const int r = div_magic(p, S.magic_L1), k = p - r * Nc;
            const cplx zk = buf[r * LP + pad_e(k == H ? 0 : k)];
            cplx zc = buf[r * LP + pad_e(k == 0 ? 0 : H - k)];
            zc.y = -zc.y;
            const cplx e = c_add(zk, zc), o = c_mul(c_sub(zk, zc), wk[i]);
            cplx x; x.x = 0.5 * (e.x + o.y); x.y = 0.5 * (e.y - o.x);             // e - i o
            FQ[(size_t)blockIdx.y * plane + ((size_t)(k >> lc) * M + y0 + r) * C + (k & (C - 1))] = x;
        }
    }
}

// Column transforms + cross power + inverse column transforms: workgroup = C = 2^lc adjacent columns of one job (blockIdx.y), all M rows.
// LDS: 2 C transforms of M points (image A's columns, then B's; padded layout).  Rows [hh, M) of the forward spectra are zero and not read.
__global__ __launch_bounds__(512) void k_phase_cols(const cplx *__restrict__ FQ, cplx *__restrict__ CP, int hh, int Nc, int N, int lc, const FftSched S,
                                                     const cplx *__restrict__ tabM)
{
    extern __shared__ __align__(16) unsigned char phase_lds[];
    const int M = S.L, LP = fft_lp(M), T = blockDim.x, tid = threadIdx.x, C = 1 << lc;
    cplx *buf = (cplx *)phase_lds;
    const int v0 = blockIdx.x * C;
    const int cols = min(C, Nc - v0);
    const size_t plane = (size_t)M * gridDim.x * C;                           // [tile][row][C]: this workgroup's tile is CONTIGUOUS
    const cplx *A = FQ + (size_t)(2 * blockIdx.y) * plane + (size_t)blockIdx.x * M * C;
    {
        cplx z[PHASE_NI];
#pragma unroll
        for (int i = 0; i < PHASE_NI; i++) {
            const int p = tid + i * T;                                           // p = (img * M + u) * C + c
            const int t = p >> lc, img = t >= M ? 1 : 0, u = min(t - img * M, hh - 1);
            z[i] = A[(size_t)img * plane + (size_t)u * C + (p & (C - 1))];
        }
#pragma unroll
        for (int i = 0; i < PHASE_NI; i++) {
            const int p = tid + i * T;
            const int c = p & (C - 1), t = p >> lc, img = t >= M ? 1 : 0, u = t - img * M;
            if (t < 2 * M) {
                cplx o = z[i];
                if (u >= hh || c >= cols) { o.x = 0.0; o.y = 0.0; }
                buf[(img * C + c) * LP + pad_e(u)] = o;
            }
        }
    }
    __syncthreads();
#if !(PHASE_DBG & 1)
    fft_lds(buf, tabM, S, LP, 2 * C, -1.0, tid, T);
#endif
    const double eps = DBL_EPSILON;
#if !(PHASE_DBG & 4)
    for (int p = tid; p < M * C; p += T) {
        const int c = div_magic(p, S.magic_L), u = p - c * M, v = v0 + c;
        const cplx f1 = buf[c * LP + pad_e(u)], f2 = buf[(C + c) * LP + pad_e(u)];
        const double pr = f1.x * f2.x + f1.y * f2.y;
        const double pi = f1.y * f2.x - f1.x * f2.y;
        const bool real_u = (u == 0) || ((M % 2 == 0) && u == M / 2);
        const bool real_v = (v == 0) || ((N % 2 == 0) && v == N / 2);
        cplx o;
        if (real_u && real_v) {
            const double mg = pr * pr;
            o.x = pr / (mg + eps); o.y = 0.0;
        } else {
            const double mg = sqrt(pr * pr + pi * pi);
            const double denom = mg * mg + eps;
            o.x = (pr * mg) / denom;
            o.y = (pi * mg) / denom;
        }
        buf[c * LP + pad_e(u)] = o;
    }
    __syncthreads();
#endif
#if !(PHASE_DBG & 2)
    fft_lds(buf, tabM, S, LP, C, 1.0, tid, T);
#endif
    cplx *O = CP + (size_t)blockIdx.y * plane + (size_t)blockIdx.x * M * C;   // the same tiled layout: one contiguous run
    for (int p = tid; p < M * C; p += T) O[p] = buf[(p & (C - 1)) * LP + pad_e(p >> lc)];
}

// Inverse row transforms + the arg-max of the row block: workgroup = RPW rows of one job (blockIdx.y).  CP: [nb][M][H + 1] -> RE: [nb][M][2 H].
// tr: the planes hold the TRANSPOSED problem (stored row = original column); the arg-max index is over the original, shifted surface.
__global__ __launch_bounds__(512) void k_phase_rows_inv(const cplx *__restrict__ CP, double *__restrict__ RE, int M, int RPW, int lc, const FftSched S,
                                                         const cplx *__restrict__ tabN, int tr, ArgMax *__restrict__ partial)
{
    extern __shared__ __align__(16) unsigned char phase_lds[];
    const int H = S.L, LP = fft_lp(H), Nc = H + 1, N = 2 * H, T = blockDim.x, tid = threadIdx.x;
    cplx *buf = (cplx *)phase_lds;
    const int y0 = blockIdx.x * RPW;
    const int rows = min(RPW, M - y0);
    const int npts = rows * H;
    const int C = 1 << lc;
    const cplx *G = CP + (size_t)blockIdx.y * M * (((Nc + C - 1) >> lc) << lc) + (size_t)y0 * C;      // [tile][row][C]
    // Z[k] = (G[k] + conj G[H-k]) + i e^(+2 pi i k / N) (G[k] - conj G[H-k]),  k = 0 .. H-1; the imaginary parts of the DC and Nyquist bins do
    // not exist in the packed format of the reference (and rocFFT's real inverse ignores them): dropped
    {
        cplx gk[PHASE_NI], gc[PHASE_NI], w[PHASE_NI];
#pragma unroll
        for (int i = 0; i < PHASE_NI; i++) {
            const int p = min(tid + i * T, npts - 1);
            const int r = div_magic(p, S.magic_L), k = p - r * H;
            const int kc = H - k;
            gk[i] = G[((size_t)(k >> lc) * M + r) * C + (k & (C - 1))]; gc[i] = G[((size_t)(kc >> lc) * M + r) * C + (kc & (C - 1))]; w[i] = tabN[k];
        }
#pragma unroll
        for (int i = 0; i < PHASE_NI; i++) {
            const int p = tid + i * T;
            if (p < npts) {
                const int r = div_magic(p, S.magic_L), k = p - r * H;
                cplx a = gk[i], b = gc[i], ww = w[i];
                if (k == 0) { a.y = 0.0; b.y = 0.0; }
                b.y = -b.y; ww.y = -ww.y;
                const cplx e = c_add(a, b), o = c_mul(c_sub(a, b), ww);
                cplx z; z.x = e.x - o.y; z.y = e.y + o.x;                          // e + i o
                buf[r * LP + pad_e(k)] = z;
            }
        }
    }
    __syncthreads();
    fft_lds(buf, tabN, S, LP, rows, 1.0, tid, T);
    ArgMax best; best.v = -INFINITY; best.idx = (long long)M * N;
    double *R = RE + ((size_t)blockIdx.y * M + y0) * N;
    const int oM = tr ? N : M, oN = tr ? M : N;                               // the original padded size
    for (int p = tid; p < npts; p += T) {
        const int r = div_magic(p, S.magic_L), n = p - r * H;
        const cplx z = buf[r * LP + pad_e(n)];
        double2 o; o.x = z.x; o.y = z.y;
        *(double2 *)(R + (size_t)r * N + 2 * n) = o;
        const int sr = y0 + r, sc = 2 * n;                                    // stored (row, column) of z.x; z.y is one column on
        ArgMax c0, c1;
        c0.v = z.x; c1.v = z.y;
        int y0s, x0s, y1s, x1s;                                                // the shift is its own inverse: source -> shifted
        if (tr) { unshift2(sc, sr, oM, oN, y0s, x0s); unshift2(sc + 1, sr, oM, oN, y1s, x1s); }
        else { unshift2(sr, sc, oM, oN, y0s, x0s); unshift2(sr, sc + 1, oM, oN, y1s, x1s); }
        c0.idx = (long long)y0s * oN + x0s;
        c1.idx = (long long)y1s * oN + x1s;
        best = better(best, better(c0, c1));
    }
    __shared__ ArgMax sm[8];
    for (int d = 32; d > 0; d >>= 1) {
        ArgMax o;
        o.v = __shfl_down(best.v, d); o.idx = __shfl_down(best.idx, d);
        best = better(best, o);
    }
    if ((tid & 63) == 0) sm[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < (T + 63) / 64; k++) best = better(best, sm[k]);
        partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = best;
    }
}

// first maximum of a job's partial records -> the 5 x 5 clamped centroid of phasecorr.cpp weightedCentroid; surface element (y, x) of the
// shifted plane = R[uy * sy + ux * sx], (uy, ux) = unshift2(y, x)
__global__ __launch_bounds__(64) void k_peak_centroid(const double *__restrict__ RE, int M, int N, long long sy, long long sx, const ArgMax *partial, int nparts,
                                                      double *out3)
{
    const double *R = RE + (size_t)blockIdx.x * M * N;
    ArgMax best; best.v = -INFINITY; best.idx = (long long)M * N;
    for (int k = threadIdx.x; k < nparts; k += 64) best = better(best, partial[(size_t)blockIdx.x * nparts + k]);
    for (int d = 32; d > 0; d >>= 1) {
        ArgMax o;
        o.v = __shfl_down(best.v, d); o.idx = __shfl_down(best.idx, d);
        best = better(best, o);
    }
    best.v = __shfl(best.v, 0); best.idx = __shfl(best.idx, 0);
    const int py = (int)(best.idx / N), px = (int)(best.idx % N);
    int minr = py - 2, maxr = py + 2, minc = px - 2, maxc = px + 2;
    if (minr < 0) minr = 0;
    if (minc < 0) minc = 0;
    if (maxr > M - 1) maxr = M - 1;
    if (maxc > N - 1) maxc = N - 1;
    // the (at most) 25 window elements are fetched by 25 lanes at once; lane 0 then accumulates them in the reference's row-major order
    const int wc = maxc - minc + 1, cnt = (maxr - minr + 1) * wc;
    double mine = 0.0;
    if ((int)threadIdx.x < cnt) {
        const int y = minr + (int)threadIdx.x / wc, x = minc + (int)threadIdx.x % wc;
        int uy, ux;
        unshift2(y, x, M, N, uy, ux);
        mine = R[(size_t)uy * sy + (size_t)ux * sx];
    }
    double cx = 0, cy = 0, s = 0;
    for (int q = 0; q < cnt; q++) {
        const double v = __shfl(mine, q);
        const int y = minr + q / wc, x = minc + q % wc;
        cx += (double)x * v; cy += (double)y * v; s += v;
    }
    if (threadIdx.x != 0) return;
    double response = s;
    s += DBL_EPSILON;
    cx /= s; cy /= s;
    response /= (double)M * (double)N;
    double *o = out3 + 3 * (size_t)blockIdx.x;
    o[0] = (double)N / 2.0 - cx;
    o[1] = (double)M / 2.0 - cy;
    o[2] = response;
}

// ---- rocFFT plans ---------------------------------------------------------------------------------------------------------------
static bool g_rocfft_ready = false;

static int get_plan(vfsms_ctx *ctx, int M, int N, int nb, FftPlan **out)
{
    for (auto &p : ctx->plans)
        if (p.M == M && p.N == N && p.nb == nb) { *out = &p; return VFSMS_OK; }
    if (!g_rocfft_ready) { if (rocfft_setup() != rocfft_status_success) { vfsms_set_error("rocfft_setup failed"); return VFSMS_ERR_FFT; } g_rocfft_ready = true; }
    FftPlan rec; memset(&rec, 0, sizeof(rec));
    rec.M = M; rec.N = N; rec.nb = nb;
    const size_t lengths[2] = {(size_t)N, (size_t)M};                 // rocFFT: fastest dimension first
    rocfft_plan f = nullptr, i = nullptr;
    rocfft_execution_info fi = nullptr, ii = nullptr;
    if (rocfft_plan_create(&f, rocfft_placement_notinplace, rocfft_transform_type_real_forward, rocfft_precision_double, 2, lengths,
                           (size_t)2 * nb, nullptr) != rocfft_status_success ||
        rocfft_plan_create(&i, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, rocfft_precision_double, 2, lengths,
                           (size_t)nb, nullptr) != rocfft_status_success) {
        vfsms_set_error("rocfft_plan_create(%d x %d, batch %d) failed", M, N, nb);
        return VFSMS_ERR_FFT;
    }
    rocfft_plan_get_work_buffer_size(f, &rec.fwd_work);
    rocfft_plan_get_work_buffer_size(i, &rec.inv_work);
    if (rocfft_execution_info_create(&fi) != rocfft_status_success || rocfft_execution_info_create(&ii) != rocfft_status_success ||
        rocfft_execution_info_set_stream(fi, ctx->stream) != rocfft_status_success ||
        rocfft_execution_info_set_stream(ii, ctx->stream) != rocfft_status_success) {
        vfsms_set_error("rocfft_execution_info setup failed"); return VFSMS_ERR_FFT;
    }
    rec.fwd = (void *)f; rec.inv = (void *)i; rec.fwd_info = (void *)fi; rec.inv_info = (void *)ii;
    ctx->plans.push_back(rec);
    *out = &ctx->plans.back();
    return VFSMS_OK;
}

int phase_destroy_plans(vfsms_ctx *ctx)
{
    for (auto &p : ctx->plans) {
        rocfft_plan_destroy((rocfft_plan)p.fwd); rocfft_plan_destroy((rocfft_plan)p.inv);
        rocfft_execution_info_destroy((rocfft_execution_info)p.fwd_info); rocfft_execution_info_destroy((rocfft_execution_info)p.inv_info);
    }
    ctx->plans.clear();
    for (auto &t : ctx->fft_tabs) hipFree(t.second);
    ctx->fft_tabs.clear();
    return VFSMS_OK;
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// rocFFT plans are cached per (M, N, batch): the registrar's speculative batches come in every size from 1 to 32, and a plan costs tens
// of milliseconds to create -- inside a timed call.  Batches are therefore rounded up to ten sizes; the surplus transforms run over
// scratch planes nobody reads (the pad / cross-power / arg-max kernels only cover the real jobs).
static inline int plan_batch(int c)
{
    static const int sizes[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
    for (int s : sizes) if (s >= c) return s;
    return c;
}

// ---- host side of the LDS transforms ---------------------------------------------------------------------------------------------------------
#define PHASE_LDS_MAX 162816      // dynamic LDS of a workgroup: the 160 KiB of a CU less 1 KiB for the static records of the kernels
struct OwnShape { int tr, M, N, H, hh, ww, RPW, Trow, C, lc, Tcol; size_t lds_row, lds_col; };

static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

static inline int round64(int x) { return std::min(512, std::max(64, (x + 63) / 64 * 64)); }

// Can the strip h x w be correlated by the LDS transforms, and in which orientation?  The shorter padded axis becomes the column axis.
static bool phase_own_shape(int h, int w, OwnShape *S)
{
    if (!env_int("VFSMS_PHASE_LDS_FFT", 1)) return false;
    const int Mo = optimal_dft_size(h), No = optimal_dft_size(w);
    int tr = Mo > No;
    for (int attempt = 0; attempt < 2; attempt++, tr = !tr) {
        const int M = tr ? No : Mo, N = tr ? Mo : No;
        if ((N & 1) || N < 4 || M < 2) continue;
        const int H = N / 2;
        if (H > 4096) continue;
        const int RPW = std::max(1, std::min(16, std::max(1, env_int("VFSMS_PHASE_ROWPTS", 2048)) / H));      // rows of a workgroup: ~2048 points
        const int tdiv = std::max(1, env_int("VFSMS_PHASE_TDIV", 8));                    // points of a workgroup per thread
        const int Trow = round64((RPW * H + tdiv - 1) / tdiv);
        const size_t lds_row = sizeof(cplx) * (size_t)RPW * fft_lp(H);
        if (RPW * H > 8 * Trow || lds_row > PHASE_LDS_MAX) continue;
        int C = env_int("VFSMS_PHASE_C", 4);
        while (C >= 2 && (sizeof(cplx) * (size_t)2 * C * fft_lp(M) > PHASE_LDS_MAX || 2 * C * M > 4096)) C >>= 1;
        if (C < 2) continue;
        int lc = 0;
        while ((1 << (lc + 1)) <= C) lc++;
        C = 1 << lc;                                                            // a power of two (shifts in the kernel)
        S->lc = lc;
        S->tr = tr; S->M = M; S->N = N; S->H = H; S->hh = tr ? w : h; S->ww = tr ? h : w; S->RPW = RPW; S->Trow = Trow; S->C = C;
        S->Tcol = round64(env_int("VFSMS_PHASE_TCOL", (2 * C * M + tdiv - 1) / tdiv));
        if (2 * C * M > 8 * S->Tcol) S->Tcol = round64((2 * C * M + 7) / 8);
        S->lds_row = lds_row; S->lds_col = sizeof(cplx) * (size_t)2 * C * fft_lp(M);
        return true;
    }
    return false;
}

static FftSched make_sched(int L, int Ltab)      // Ltab: length of the twiddle table the passes index (L, or 2 L for the packed rows)
{
    FftSched S; memset(&S, 0, sizeof(S));
    S.L = L;
    S.magic_L = L > 1 ? 0xFFFFFFFFu / (unsigned)L + 1u : 0u;
    S.magic_L1 = 0xFFFFFFFFu / (unsigned)(L + 1) + 1u;
    int rest = L, Ls = 1;
    auto push = [&](int R) {
        FftPass &P = S.p[S.n++];
        P.R = R; P.Ls = Ls; P.m = L / R; P.twstride = Ltab / (Ls * R);
        P.magic_m = P.m > 1 ? 0xFFFFFFFFu / (unsigned)P.m + 1u : 0u;
        P.magic_Ls = Ls > 1 ? 0xFFFFFFFFu / (unsigned)Ls + 1u : 0u;
        Ls *= R; rest /= R;
    };
    // the largest in-register butterflies first (16 with Ls = 1 needs no twiddles at all): 432 = 16 x 9 x 3, 1024 = 16 x 16 x 4, 864 = 16 x 6 x 9
    int a = 0, b = 0, c = 0;
    while (rest % 2 == 0) { a++; rest /= 2; }
    while (rest % 3 == 0) { b++; rest /= 3; }
    while (rest % 5 == 0) { c++; rest /= 5; }
    rest = L;
    for (; a >= 4; a -= 4) push(16);
    if (a == 3) push(8);
    if (a == 2) push(4);
    if (a == 1) {
        if (b >= 1) { push(6); b--; }
        else if (c >= 1) { push(10); c--; }
        else push(2);
    }
    for (; b >= 2; b -= 2) push(9);
    if (b) push(3);
    for (; c > 0; c--) push(5);
    return S;
}

// exp(-2 pi i q / L), q = 0 .. L-1, rounded from long double; cached per length for the life of the context
static int get_tab(vfsms_ctx *ctx, int L, const cplx **out)
{
    for (auto &t : ctx->fft_tabs) if (t.first == L) { *out = (const cplx *)t.second; return VFSMS_OK; }
    std::vector<cplx> h(L);
    const long double tau = 6.283185307179586476925286766559005768L;
    for (int q = 0; q < L; q++) {
        // exact symmetries first, so that the quarter points are exact
        const long double a = tau * (long double)q / (long double)L;
        h[q].x = (double)cosl(a); h[q].y = (double)-sinl(a);
        if (4 * q == L) { h[q].x = 0.0; h[q].y = -1.0; }
        if (2 * q == L) { h[q].x = -1.0; h[q].y = 0.0; }
        if (4 * q == 3 * L) { h[q].x = 0.0; h[q].y = 1.0; }
    }
    h[0].x = 1.0; h[0].y = 0.0;
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(cplx) * L));
    HIP_TRY(hipMemcpy(d, h.data(), sizeof(cplx) * L, hipMemcpyHostToDevice));
    ctx->fft_tabs.push_back({L, d});
    *out = (const cplx *)d;
    return VFSMS_OK;
}

extern "C" int vfsms_phase_plan(int h, int w, int32_t *info8)
{
    if (h <= 0 || w <= 0 || !info8) { vfsms_set_error("phase_plan: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    OwnShape S;
    memset(info8, 0, sizeof(int32_t) * 8);
    if (phase_own_shape(h, w, &S)) {
        const int32_t v[8] = {1, S.tr, S.M, S.N, S.C, S.RPW, S.Trow, S.Tcol};
        memcpy(info8, v, sizeof(v));
    } else {
        info8[2] = optimal_dft_size(h); info8[3] = optimal_dft_size(w);
    }
    return VFSMS_OK;
}

static size_t own_bytes(const OwnShape &S, int h, int w, int nb)
{
    const size_t c = (size_t)std::min(nb, PHASE_MAX_CHUNK);
    const size_t cpl = sizeof(cplx) * (size_t)S.M * (((S.H + 1 + S.C - 1) / S.C) * S.C);
    return al256(c * sizeof(double) * S.M * S.N) + al256(2 * c * cpl) + al256(c * cpl) + al256(2 * c * (size_t)h * w) + al256(sizeof(PhaseJob) * 2 * nb) +
           al256(sizeof(ArgMax) * c * S.M) + 65536;
}

static int phase_own_batch(vfsms_ctx *ctx, const OwnShape &S, const PhaseJobHost *jobs, int nb, int h, int w, double *d_out3)
{
    if (ctx->fft_tabs.empty()) {                               // first use on this context (= this device): > 64 KB of dynamic LDS must be asked for
        HIP_TRY(hipFuncSetAttribute((const void *)k_phase_rows_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, PHASE_LDS_MAX));
        HIP_TRY(hipFuncSetAttribute((const void *)k_phase_cols, hipFuncAttributeMaxDynamicSharedMemorySize, PHASE_LDS_MAX));
        HIP_TRY(hipFuncSetAttribute((const void *)k_phase_rows_inv, hipFuncAttributeMaxDynamicSharedMemorySize, PHASE_LDS_MAX));
    }
    const int M = S.M, N = S.N, Nc = S.H + 1;
    const int cmax = std::min(nb, std::max(1, std::min(PHASE_MAX_CHUNK, env_int("VFSMS_PHASE_CHUNK", PHASE_MAX_CHUNK))));
    const size_t real = (size_t)M * N, cpl = (size_t)M * (((Nc + S.C - 1) / S.C) * S.C);          // spectra in column tiles of C
    const int nparts = (M + S.RPW - 1) / S.RPW;
    double *RE = (double *)ctx_arena_alloc(ctx, sizeof(double) * cmax * real);
    cplx *FQ = (cplx *)ctx_arena_alloc(ctx, sizeof(cplx) * 2 * cmax * cpl);
    cplx *CP = (cplx *)ctx_arena_alloc(ctx, sizeof(cplx) * cmax * cpl);
    uint8_t *TB = S.tr ? (uint8_t *)ctx_arena_alloc(ctx, 2 * (size_t)cmax * h * w) : nullptr;
    ArgMax *partial = (ArgMax *)ctx_arena_alloc(ctx, sizeof(ArgMax) * cmax * nparts);
    if (!RE || !FQ || !CP || !partial || (S.tr && !TB)) { vfsms_set_error("arena exhausted in phase correlation"); return VFSMS_ERR_CAPACITY; }
    // job records: the caller's strips, then (transposed orientation) the scratch copies the row kernel reads instead, chunk-relative
    std::vector<PhaseJob> hj((size_t)nb + (S.tr ? cmax : 0));
    for (int k = 0; k < nb; k++) { hj[k].a = jobs[k].a; hj[k].b = jobs[k].b; hj[k].sa = jobs[k].sa; hj[k].sb = jobs[k].sb; }
    for (int k = 0; S.tr && k < cmax; k++) {
        hj[nb + k].a = TB + (size_t)(2 * k) * h * w; hj[nb + k].b = TB + (size_t)(2 * k + 1) * h * w; hj[nb + k].sa = h; hj[nb + k].sb = h;
    }
    PhaseJob *dj = nullptr;
    TRY(ctx_upload_small(ctx, hj.data(), sizeof(PhaseJob) * hj.size(), (void **)&dj));
    const cplx *tabN = nullptr, *tabM = nullptr;
    TRY(get_tab(ctx, N, &tabN));
    TRY(get_tab(ctx, M, &tabM));
    const FftSched SR = make_sched(S.H, N), SC = make_sched(M, M);
    ProfScope ps(ctx, "phase");
    for (int done = 0; done < nb;) {
        const int c = std::min(cmax, nb - done);
        const PhaseJob *rowjobs = dj + done;
        if (S.tr) {
            hipLaunchKernelGGL(k_phase_transpose_u8, dim3((w + 63) / 64, (h + 63) / 64, 2 * c), dim3(256), 0, ctx->stream, dj + done, h, w, TB);
            rowjobs = dj + nb;
        }
        hipLaunchKernelGGL(k_phase_rows_fwd, dim3((S.hh + S.RPW - 1) / S.RPW, 2 * c), dim3(S.Trow), S.lds_row, ctx->stream, rowjobs, S.hh, S.ww, M, S.RPW, S.lc,
                           SR, tabN, FQ);
        hipLaunchKernelGGL(k_phase_cols, dim3((Nc + S.C - 1) / S.C, c), dim3(S.Tcol), S.lds_col, ctx->stream, (const cplx *)FQ, CP, S.hh, Nc, N, S.lc, SC, tabM);
        hipLaunchKernelGGL(k_phase_rows_inv, dim3(nparts, c), dim3(S.Trow), S.lds_row, ctx->stream, (const cplx *)CP, RE, M, S.RPW, S.lc, SR, tabN, S.tr, partial);
        // the surface as the reference indexes it: oM x oN (the original padded size)
        const int oM = S.tr ? N : M, oN = S.tr ? M : N;
        hipLaunchKernelGGL(k_peak_centroid, dim3(c), dim3(64), 0, ctx->stream, (const double *)RE, oM, oN, (long long)(S.tr ? 1 : N), (long long)(S.tr ? N : 1),
                           (const ArgMax *)partial, nparts, d_out3 + 3 * (size_t)done);
        done += c;
    }
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

// arena bytes for nb attempts of one ROI size; the rocFFT work buffer is whatever the (cached) plans of the chunks ask for
int phase_bytes(vfsms_ctx *ctx, int h, int w, int nb, size_t *bytes)
{
    OwnShape own;
    if (phase_own_shape(h, w, &own)) { *bytes = own_bytes(own, h, w, nb); return VFSMS_OK; }
    const int M = optimal_dft_size(h), N = optimal_dft_size(w);
    const size_t real = sizeof(double) * (size_t)M * N, cp = sizeof(double) * 2 * (size_t)M * (N / 2 + 1);
    const size_t chunk = (size_t)plan_batch(std::min(nb, PHASE_MAX_CHUNK));
    size_t wbytes = 0;
    for (int left = nb; left > 0;) {
        const int c = std::min(left, PHASE_MAX_CHUNK);
        FftPlan *P;
        TRY(get_plan(ctx, M, N, plan_batch(c), &P));
        wbytes = std::max(wbytes, std::max(P->fwd_work, P->inv_work));
        left -= c;
    }
    *bytes = al256(2 * chunk * real) + al256(2 * chunk * cp) + al256(chunk * cp) + al256(wbytes) +
             al256(sizeof(PhaseJob) * nb) + al256(sizeof(ArgMax) * PHASE_NBLK * chunk) + 65536;
    return VFSMS_OK;
}

// nb attempts of one ROI size h x w.  jobs: HOST array of device pointers / strides.  d_out3: device double[3 * nb].
// Scratch comes from the context arena (caller reserved phase_bytes(h, w, nb)).  Stream-ordered, no host sync.
int phase_correlate_batch_device(vfsms_ctx *ctx, const PhaseJobHost *jobs, int nb, int h, int w, double *d_out3)
{
    if (nb <= 0) return VFSMS_OK;
    OwnShape own;
    if (phase_own_shape(h, w, &own)) return phase_own_batch(ctx, own, jobs, nb, h, w, d_out3);
    const int M = optimal_dft_size(h), N = optimal_dft_size(w);
    const int Nc = N / 2 + 1;
    const size_t real = (size_t)M * N, cpl = (size_t)M * Nc;
    const int cmax = plan_batch(std::min(nb, PHASE_MAX_CHUNK));
    double *RE = (double *)ctx_arena_alloc(ctx, sizeof(double) * 2 * cmax * real);
    cplx *FQ = (cplx *)ctx_arena_alloc(ctx, sizeof(cplx) * 2 * cmax * cpl);
    cplx *CP = (cplx *)ctx_arena_alloc(ctx, sizeof(cplx) * cmax * cpl);
    ArgMax *partial = (ArgMax *)ctx_arena_alloc(ctx, sizeof(ArgMax) * PHASE_NBLK * cmax);
    std::vector<PhaseJob> hj(nb);
    for (int k = 0; k < nb; k++) { hj[k].a = jobs[k].a; hj[k].b = jobs[k].b; hj[k].sa = jobs[k].sa; hj[k].sb = jobs[k].sb; }
    PhaseJob *dj = nullptr;
    TRY(ctx_upload_small(ctx, hj.data(), sizeof(PhaseJob) * nb, (void **)&dj));
    if (!RE || !FQ || !CP || !partial) { vfsms_set_error("arena exhausted in phase correlation"); return VFSMS_ERR_CAPACITY; }
    // chunks of at most PHASE_MAX_CHUNK attempts: plans first, so one work buffer serves every chunk
    std::vector<FftPlan *> chunks;
    std::vector<int> chunk_jobs;
    size_t wbytes = 0;
    for (int left = nb; left > 0;) {
        const int c = std::min(left, PHASE_MAX_CHUNK);
        FftPlan *P;
        TRY(get_plan(ctx, M, N, plan_batch(c), &P));
        chunks.push_back(P); chunk_jobs.push_back(c);
        wbytes = std::max(wbytes, std::max(P->fwd_work, P->inv_work));
        left -= c;
    }
    void *work = wbytes ? ctx_arena_alloc(ctx, wbytes) : nullptr;
    if (wbytes && !work) { vfsms_set_error("arena exhausted (rocFFT work buffer, %zu bytes)", wbytes); return VFSMS_ERR_CAPACITY; }
    ProfScope ps(ctx, "phase");
    int done = 0;
    for (size_t ci = 0; ci < chunks.size(); ci++) {
        FftPlan *P = chunks[ci];
        const int c = chunk_jobs[ci];                                   // real jobs; the plan transforms P->nb >= c planes
        hipLaunchKernelGGL(k_pad_u8_f64, dim3((N + 255) / 256, M, c), dim3(256), 0, ctx->stream, dj + done, h, w, M, N, RE);
        void *in[1] = {RE}, *outb[1] = {FQ};
        if (wbytes) rocfft_execution_info_set_work_buffer((rocfft_execution_info)P->fwd_info, work, wbytes);
        if (rocfft_execute((rocfft_plan)P->fwd, in, outb, (rocfft_execution_info)P->fwd_info) != rocfft_status_success) {
            vfsms_set_error("rocfft_execute (forward) failed"); return VFSMS_ERR_FFT;
        }
        hipLaunchKernelGGL(k_cross_power, dim3((Nc + 255) / 256, M, c), dim3(256), 0, ctx->stream, FQ, CP, M, N);
        void *in2[1] = {CP}, *out2[1] = {RE};
        if (wbytes) rocfft_execution_info_set_work_buffer((rocfft_execution_info)P->inv_info, work, wbytes);
        if (rocfft_execute((rocfft_plan)P->inv, in2, out2, (rocfft_execution_info)P->inv_info) != rocfft_status_success) {
            vfsms_set_error("rocfft_execute (inverse) failed"); return VFSMS_ERR_FFT;
        }
        hipLaunchKernelGGL(k_argmax_partial, dim3(PHASE_NBLK, c), dim3(256), 0, ctx->stream, RE, M, N, partial);
        hipLaunchKernelGGL(k_argmax_centroid, dim3(c), dim3(64), 0, ctx->stream, RE, M, N, partial, d_out3 + 3 * (size_t)done);
        done += c;                                                      // the scratch planes are reused by the next chunk (stream order)
    }
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

int phase_correlate_device(vfsms_ctx *ctx, const uint8_t *a, int stride_a, const uint8_t *b, int stride_b,
                           int h, int w, double *d_out3)
{
    PhaseJobHost j; j.a = a; j.b = b; j.sa = stride_a; j.sb = stride_b;
    return phase_correlate_batch_device(ctx, &j, 1, h, w, d_out3);
}

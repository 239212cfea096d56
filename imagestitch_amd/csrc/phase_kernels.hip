// phase_kernels.hip -- FP64 phase correlation for gfx950: hipFFT (rocFFT backend) D2Z / Z2D plans cached
// per padded size, with hand-written pad / cross-power / shifted-argmax / centroid kernels around them.
//
// Replaces cv2.phaseCorrelate(np.float64(roiA), np.float64(roiB)) at Stitcher.py:230, i.e. OpenCV 3.3.1
// imgproc/src/phasecorr.cpp semantics: zero-pad bottom/right to
// getOptimalDFTSize, P = F1 * conj(F2), C = idft(P / |P|) unscaled with the packed-format quirk that the
// purely-real bins (DC / Nyquist) divide by x*x instead of |x|, fftShift by quadrant swap (odd sizes keep
// their last row/column), first-maximum argmax, 5x5 clamped weighted centroid, response / (M*N).
// All of it is HBM-bound streaming work; the u8 -> f64 conversion is fused into the pad kernel and the
// quadrant swap is never materialised (the argmax and centroid kernels index through it).
#include "common.h"
#include <hipfft/hipfft.h>
#include <float.h>

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

__global__ __launch_bounds__(256) void k_pad_u8_f64(const uint8_t *__restrict__ a, int sa, const uint8_t *__restrict__ b, int sb,
                                                    int h, int w, int M, int N, double *__restrict__ A, double *__restrict__ B)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= N) return;
    const bool in = (y < h) && (x < w);
    const size_t o = (size_t)y * N + x;
    A[o] = in ? (double)a[(size_t)y * sa + x] : 0.0;
    B[o] = in ? (double)b[(size_t)y * sb + x] : 0.0;
}

// mulSpectrums(conjB) + magSpectrums + divSpectrums on the half spectrum (M x (N/2+1) complex)
__global__ __launch_bounds__(256) void k_cross_power(hipfftDoubleComplex *__restrict__ F1, const hipfftDoubleComplex *__restrict__ F2,
                                                     int M, int N)
{
    const int Nc = N / 2 + 1;
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int u = blockIdx.y;
    if (v >= Nc) return;
    const size_t k = (size_t)u * Nc + v;
    const double ar = F1[k].x, ai = F1[k].y, br = F2[k].x, bi = F2[k].y;
    const double pr = ar * br + ai * bi;
    const double pi = ai * br - ar * bi;
    const bool real_u = (u == 0) || ((M % 2 == 0) && u == M / 2);
    const bool real_v = (v == 0) || ((N % 2 == 0) && v == N / 2);
    const double eps = DBL_EPSILON;
    hipfftDoubleComplex c;
    if (real_u && real_v) {
        const double mg = pr * pr;
        c.x = pr / (mg + eps); c.y = 0.0;
    } else {
        const double mg = sqrt(pr * pr + pi * pi);
        const double denom = mg * mg + eps;
        c.x = (pr * mg) / denom;
        c.y = (pi * mg) / denom;
    }
    F1[k] = c;
}

// shifted coordinate -> source coordinate of phasecorr.cpp fftShift (quadrant swap of size n>>1)
__device__ __forceinline__ int unshift(int s, int n)
{
    const int mid = n >> 1;
    if (s < mid) return s + mid;
    if (s < 2 * mid) return s - mid;
    return s;                                           // odd n: last row / column stays in place
}

struct ArgMax { double v; long long idx; };

__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b)
{
    // larger value wins; equal values -> smaller shifted index (minMaxLoc returns the first maximum)
    if (b.v > a.v || (b.v == a.v && b.idx < a.idx)) return b;
    return a;
}

__global__ __launch_bounds__(256) void k_argmax_partial(const double *__restrict__ R, int M, int N, ArgMax *partial)
{
    const long long total = (long long)M * N;
    ArgMax best; best.v = -INFINITY; best.idx = total;
    for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < total; s += (long long)gridDim.x * 256) {
        const int ys = (int)(s / N), xs = (int)(s % N);
        const double v = R[(size_t)unshift(ys, M) * N + unshift(xs, N)];
        ArgMax c; c.v = v; c.idx = s;
        best = better(best, c);
    }
    __shared__ ArgMax sm[256];
    sm[threadIdx.x] = best;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) sm[threadIdx.x] = better(sm[threadIdx.x], sm[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

__global__ __launch_bounds__(256) void k_argmax_centroid(const double *__restrict__ R, int M, int N, const ArgMax *partial,
                                                         int npartial, double *out3)
{
    __shared__ ArgMax sm[256];
    ArgMax best; best.v = -INFINITY; best.idx = (long long)M * N;
    for (int k = threadIdx.x; k < npartial; k += 256) best = better(best, partial[k]);
    sm[threadIdx.x] = best;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
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
            const double v = R[(size_t)unshift(y, M) * N + unshift(x, N)];
            cx += (double)x * v; cy += (double)y * v; s += v;
        }
    double response = s;
    s += DBL_EPSILON;
    cx /= s; cy /= s;
    response /= (double)M * (double)N;
    out3[0] = (double)N / 2.0 - cx;
    out3[1] = (double)M / 2.0 - cy;
    out3[2] = response;
}

static int get_plan(vfsms_ctx *ctx, int M, int N, hipfftHandle *fwd, hipfftHandle *inv)
{
    for (auto &p : ctx->plans)
        if (p.M == M && p.N == N) { *fwd = (hipfftHandle)p.fwd; *inv = (hipfftHandle)p.inv; return VFSMS_OK; }
    hipfftHandle f, i;
    if (hipfftPlan2d(&f, M, N, HIPFFT_D2Z) != HIPFFT_SUCCESS || hipfftPlan2d(&i, M, N, HIPFFT_Z2D) != HIPFFT_SUCCESS) {
        vfsms_set_error("hipfftPlan2d(%d, %d) failed", M, N);
        return VFSMS_ERR_FFT;
    }
    hipfftSetStream(f, ctx->stream);
    hipfftSetStream(i, ctx->stream);
    FftPlan rec; rec.M = M; rec.N = N; rec.fwd = (void *)f; rec.inv = (void *)i;
    ctx->plans.push_back(rec);
    *fwd = f; *inv = i;
    return VFSMS_OK;
}

int phase_destroy_plans(vfsms_ctx *ctx)
{
    for (auto &p : ctx->plans) { hipfftDestroy((hipfftHandle)p.fwd); hipfftDestroy((hipfftHandle)p.inv); }
    ctx->plans.clear();
    return VFSMS_OK;
}

size_t phase_bytes(int h, int w)
{
    const int M = optimal_dft_size(h), N = optimal_dft_size(w);
    const size_t real = sizeof(double) * (size_t)M * N, cplx = sizeof(double) * 2 * (size_t)M * (N / 2 + 1);
    return 2 * (real + 256) + 2 * (cplx + 256) + 4096 + 1024 * sizeof(ArgMax);
}

// a, b: device pointers to u8 ROIs.  d_out3: device double[3].  Scratch comes from the context arena
// (caller has reserved phase_bytes()).  Stream-ordered, no host sync.
int phase_correlate_device(vfsms_ctx *ctx, const uint8_t *a, int stride_a, const uint8_t *b, int stride_b,
                           int h, int w, double *d_out3)
{
    const int M = optimal_dft_size(h), N = optimal_dft_size(w);
    const int Nc = N / 2 + 1;
    hipfftHandle fwd, inv;
    TRY(get_plan(ctx, M, N, &fwd, &inv));
    double *A = (double *)ctx_arena_alloc(ctx, sizeof(double) * (size_t)M * N);
    double *B = (double *)ctx_arena_alloc(ctx, sizeof(double) * (size_t)M * N);
    hipfftDoubleComplex *F1 = (hipfftDoubleComplex *)ctx_arena_alloc(ctx, sizeof(hipfftDoubleComplex) * (size_t)M * Nc);
    hipfftDoubleComplex *F2 = (hipfftDoubleComplex *)ctx_arena_alloc(ctx, sizeof(hipfftDoubleComplex) * (size_t)M * Nc);
    const int nblk = 1024;
    ArgMax *partial = (ArgMax *)ctx_arena_alloc(ctx, sizeof(ArgMax) * nblk);
    if (!partial) { vfsms_set_error("arena exhausted in phase correlation"); return VFSMS_ERR_CAPACITY; }
    ProfScope ps(ctx, "phase");
    hipLaunchKernelGGL(k_pad_u8_f64, dim3((N + 255) / 256, M), dim3(256), 0, ctx->stream, a, stride_a, b, stride_b, h, w, M, N, A, B);
    if (hipfftExecD2Z(fwd, A, F1) != HIPFFT_SUCCESS || hipfftExecD2Z(fwd, B, F2) != HIPFFT_SUCCESS) {
        vfsms_set_error("hipfftExecD2Z failed"); return VFSMS_ERR_FFT;
    }
    hipLaunchKernelGGL(k_cross_power, dim3((Nc + 255) / 256, M), dim3(256), 0, ctx->stream, F1, F2, M, N);
    if (hipfftExecZ2D(inv, F1, A) != HIPFFT_SUCCESS) { vfsms_set_error("hipfftExecZ2D failed"); return VFSMS_ERR_FFT; }
    hipLaunchKernelGGL(k_argmax_partial, dim3(nblk), dim3(256), 0, ctx->stream, A, M, N, partial);
    hipLaunchKernelGGL(k_argmax_centroid, dim3(1), dim3(256), 0, ctx->stream, A, M, N, partial, nblk, d_out3);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

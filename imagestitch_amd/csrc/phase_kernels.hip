// phase_kernels.hip -- FP64 phase correlation for gfx950: native rocFFT real-forward / real-inverse plans, BATCHED over the
// attempts of a launch (one plan execution transforms every ROI of a batch), with hand-written batched pad / cross-power /
// shifted-argmax / centroid kernels around them.
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

#define PHASE_NBLK 64       // partial argmax blocks per job (each walks whole rows: no per-element division)

__global__ __launch_bounds__(256) void k_argmax_partial(const double *__restrict__ RE, int M, int N, ArgMax *partial)
{
    const double *R = RE + (size_t)blockIdx.y * M * N;
    ArgMax best; best.v = -INFINITY; best.idx = (long long)M * N;
    for (int ys = blockIdx.x; ys < M; ys += PHASE_NBLK) {
        const double *row = R + (size_t)unshift(ys, M) * N;
        for (int xs = threadIdx.x; xs < N; xs += 256) {
            ArgMax c; c.v = row[unshift(xs, N)]; c.idx = (long long)ys * N + xs;
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
            const double v = R[(size_t)unshift(y, M) * N + unshift(x, N)];
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

// arena bytes for nb attempts of one ROI size; the rocFFT work buffer is whatever the (cached) plans of the chunks ask for
int phase_bytes(vfsms_ctx *ctx, int h, int w, int nb, size_t *bytes)
{
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

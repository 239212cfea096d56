// match_kernels.hip -- brute-force matchers + ratio test + mode vote for gfx950.
//
// Replaces BFMatcher("BruteForce").knnMatch(k=2) + ratio filter (ImageUtility.py:288-296; DLL twin
// appendix/myGpuFeatures.cpp:160-173), BFMatcher("BruteForce-Hamming").match (ImageUtility.py:297-302)
// and Method.getOffsetByMode (ImageUtility.py:139-178).
//
// BF-L2 layout (CDNA4-native, not a warp-tiled GEMM): every lane OWNS two query descriptors in VGPRs
// (2 x 64 floats, packed so the inner loop is v_pk_add/v_pk_mul on both queries at once); train
// descriptors are wave-uniform and stream through the scalar data path (s_load) -- no LDS traffic, no
// cross-lane reduction in the hot loop: each lane keeps its own running best / second best.  Trains are
// split across blockIdx.y so that a single ROI pair still fills 256 CUs; the partial 2-NN lists are
// merged (ascending train order, ties keep the lower index) by k_merge_ratio.
// Distances: float accumulation in the 4-wide order of normL2Sqr_, sqrt per candidate exactly when the
// squared distance can still change the top-2 (comparisons happen in the sqrt domain like OpenCV's).
#include "common.h"
#include <algorithm>
#include <stdlib.h>
#include <math.h>

typedef float float2v __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(4))) cfloat;   // AMDGPU constant address space
typedef float f16v __attribute__((ext_vector_type(16)));

// hand-placed scalar loads (hipcc adds no waits for asm, cdna_hip_programming.md section 5.7): the wait takes the
// destination tuple as an in/out operand so every use of the data is ordered behind it.
#define SLOAD16(dst, ptr) asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(dst) : "s"(ptr))
// prefetch into dst while `live` (the tuple about to be consumed) is threaded through, which pins the issue point
// of the load in front of the VALU work on `live`
#define SLOAD16_BEFORE(dst, ptr, live) asm volatile("s_load_dwordx16 %0, %2, 0x0" : "=s"(dst), "+s"(live) : "s"(ptr))
#define SWAIT(dst) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dst))
#define SWAIT2(d0, d1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d0), "+s"(d1))
#define BF_CHUNK(buf, base)                                                                    \
    _Pragma("unroll") for (int d = 0; d < 16; d += 4) {                                        \
        float2v e0 = qv[base + d + 0] - buf[d + 0];                                            \
        float2v e1 = qv[base + d + 1] - buf[d + 1];                                            \
        float2v e2 = qv[base + d + 2] - buf[d + 2];                                            \
        float2v e3 = qv[base + d + 3] - buf[d + 3];                                            \
        acc += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;                                          \
    }

__device__ __forceinline__ void knn_update(float dsq, int j, float &b1, float &b2, int &i1, float &b1sq, float &b2sq)
{
    if (dsq < b2sq) {
        float d = sqrtf(dsq);
        if (d < b1) { b2 = b1; b2sq = b1sq; b1 = d; b1sq = dsq; i1 = j; }
        else if (d < b2) { b2 = d; b2sq = dsq; }
    }
}

// DIM = 64: two queries per lane (packed).  block = 256 threads = 4 waves x 128 queries.
__global__ __launch_bounds__(256) void k_bf_l2_d64(const MatchDev *jobs)
{
    const MatchDev &J = jobs[blockIdx.z];
    // device-side counts are wave-uniform: pin them to SGPRs so loop bounds and train addresses stay scalar
    const int nq = __builtin_amdgcn_readfirstlane(*J.nq_ptr), nt = __builtin_amdgcn_readfirstlane(*J.nt_ptr);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = blockIdx.x * 512 + wave * 128;
    if (q0 >= nq) return;
    const int nsplit = gridDim.y, sp = blockIdx.y;
    const int chunk = (nt + nsplit - 1) / nsplit;
    const int t0 = sp * chunk, t1 = min(nt, t0 + chunk);
    const int qa = q0 + lane, qb = q0 + 64 + lane;
    const float *__restrict__ pa = J.q + (size_t)min(qa, nq - 1) * 64;
    const float *__restrict__ pb = J.q + (size_t)min(qb, nq - 1) * 64;
    float2v qv[64];
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
        float4 va = *reinterpret_cast<const float4 *>(pa + d);
        float4 vb = *reinterpret_cast<const float4 *>(pb + d);
        qv[d + 0] = (float2v){va.x, vb.x}; qv[d + 1] = (float2v){va.y, vb.y};
        qv[d + 2] = (float2v){va.z, vb.z}; qv[d + 3] = (float2v){va.w, vb.w};
    }
    float b1a = INFINITY, b2a = INFINITY, b1sa = INFINITY, b2sa = INFINITY; int i1a = -1;
    float b1b = INFINITY, b2b = INFINITY, b1sb = INFINITY, b2sb = INFINITY; int i1b = -1;
    // Train descriptors were written by an earlier kernel and are read-only here and wave-uniform: they are
    // streamed through the scalar data path, 16 floats (one s_load_dwordx16) at a time, double-buffered by hand:
    // the load of chunk c+1 is issued before the 48 packed VALU ops on chunk c, so its latency is covered.
    // (hipcc places s_waitcnt lgkmcnt(0) right behind every scalar load it schedules itself; SMEM returns out
    // of order, so the only safe wait is 0 and it must sit BEFORE the next prefetch is issued.)
    const float *T = J.t;
    if (t0 < t1) {
        // half-train granularity: 2 x s_load_dwordx16 per wait, 96 packed VALU ops of cover for each prefetch
        f16v a0, a1, b0, b1;
        const float *first = T + (size_t)t0 * 64;
        SLOAD16(a0, first); SLOAD16(a1, first + 16);
        for (int j = t0; j < t1; j++) {
            const float *tn = T + (size_t)min(j + 1, t1 - 1) * 64;      // next train (clamped: harmless re-read)
            float2v acc = (float2v){0.f, 0.f};
            SWAIT2(a0, a1);
            SLOAD16_BEFORE(b0, T + (size_t)j * 64 + 32, a0); SLOAD16_BEFORE(b1, T + (size_t)j * 64 + 48, a1);
            BF_CHUNK(a0, 0);
            BF_CHUNK(a1, 16);
            SWAIT2(b0, b1);
            SLOAD16_BEFORE(a0, tn, b0); SLOAD16_BEFORE(a1, tn + 16, b1);
            BF_CHUNK(b0, 32);
            BF_CHUNK(b1, 48);
            knn_update(acc.x, j, b1a, b2a, i1a, b1sa, b2sa);
            knn_update(acc.y, j, b1b, b2b, i1b, b1sb, b2sb);
        }
        SWAIT2(a0, a1);                                                  // drain the last prefetch
    }
    const size_t o = (size_t)sp * J.capq;
    if (qa < nq) { J.p_d1[o + qa] = b1a; J.p_d2[o + qa] = b2a; J.p_i1[o + qa] = i1a; }
    if (qb < nq) { J.p_d1[o + qb] = b1b; J.p_d2[o + qb] = b2b; J.p_i1[o + qb] = i1b; }
}

// generic DIM (multiple of 4, <= 128): one query per lane.  block = 256 threads = 256 queries.
template <int DIM>
__global__ __launch_bounds__(256) void k_bf_l2_gen(const MatchDev *jobs)
{
    const MatchDev &J = jobs[blockIdx.z];
    // device-side counts are wave-uniform: pin them to SGPRs so loop bounds and train addresses stay scalar
    const int nq = __builtin_amdgcn_readfirstlane(*J.nq_ptr), nt = __builtin_amdgcn_readfirstlane(*J.nt_ptr);
    const int q0 = blockIdx.x * 256;
    if (q0 >= nq) return;
    const int nsplit = gridDim.y, sp = blockIdx.y;
    const int chunk = (nt + nsplit - 1) / nsplit;
    const int t0 = sp * chunk, t1 = min(nt, t0 + chunk);
    const int qa = q0 + threadIdx.x;
    const float *__restrict__ pa = J.q + (size_t)min(qa, nq - 1) * DIM;
    float qv[DIM];
#pragma unroll
    for (int d = 0; d < DIM; d++) qv[d] = pa[d];
    float b1 = INFINITY, b2 = INFINITY, b1s = INFINITY, b2s = INFINITY; int i1 = -1;
    const float *__restrict__ T = J.t;
    for (int j = t0; j < t1; j++) {
        const float *__restrict__ tr = T + (size_t)j * DIM;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < DIM; d += 4) {
            float e0 = qv[d] - tr[d], e1 = qv[d + 1] - tr[d + 1], e2 = qv[d + 2] - tr[d + 2], e3 = qv[d + 3] - tr[d + 3];
            acc += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
        }
        knn_update(acc, j, b1, b2, i1, b1s, b2s);
    }
    const size_t o = (size_t)sp * J.capq;
    if (qa < nq) { J.p_d1[o + qa] = b1; J.p_d2[o + qa] = b2; J.p_i1[o + qa] = i1; }
}

// ---------------------------------------------------------------------------------------------------
// MFMA candidate filter + exact verification (DIM = 64, descriptors of norm <= 1: the fused SURF attempt path).
//
// The 2-NN of a query under the reference's float arithmetic are, with a wide safety margin, among the trains whose
// f32-MFMA score  s(q,t) = |t|^2 - 2 q.t  (= d^2 - |q|^2 up to ~1e-5) lies within BFM_MARGIN of the second best
// score.  k_bf_mfma_d64 streams 32-train tiles against 64 resident queries per wave on v_mfma_f32_32x32x2_f32
// (K = 64 + one augmented step that adds |t|^2, so the accumulator IS the score) and appends to a small per-lane list
// every train whose score is within the margin of the lane's RUNNING second best -- a superset of the final
// candidates, since a running second best over a prefix of half the trains can only be larger than the final one.
// k_bf_verify_d64 then evaluates exactly those few trains (~10 per list) with the arithmetic of k_bf_l2_d64 (4-wide
// float accumulation, sqrt-domain compares, ties keep the lower index) -- the MFMA decides which distances get
// computed, never what they are.  A list that overflows makes its query fall back to the exhaustive exact scan.
// ---------------------------------------------------------------------------------------------------
#define BFM_CAPL 32
#define BFM_MARGIN 1e-3f        // >= 40x the worst-case |score - (d_ref^2 - |q|^2)| for K = 65, |q|,|t| <= 1
#define BFM_FAR 1e30f

struct BfmState { float m1, m2, thr; int cnt; };

__device__ __forceinline__ void bfm_warm(const f16v &acc, BfmState &S)
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float v = acc[i];
        S.m2 = fminf(S.m2, fmaxf(S.m1, v));
        S.m1 = fminf(S.m1, v);
    }
}
__device__ __forceinline__ void bfm_scan(const f16v &acc, BfmState &S, int row0, uint2 *list, size_t pitch, bool update)
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float v = acc[i];
        if (v <= S.thr && v < 1e29f) {
            if (S.cnt < BFM_CAPL) list[S.cnt * pitch] = make_uint2(__float_as_uint(v), (unsigned)(row0 + (i & 3) + 8 * (i >> 2)));
            S.cnt++;
            if (update) {
                S.m2 = fminf(S.m2, fmaxf(S.m1, v));
                S.m1 = fminf(S.m1, v);
                S.thr = S.m2 + BFM_MARGIN;
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void k_bf_mfma_d64(const MatchDev *jobs)
{
    const MatchDev &J = jobs[blockIdx.z];
    const int nq = __builtin_amdgcn_readfirstlane(*J.nq_ptr), nt = __builtin_amdgcn_readfirstlane(*J.nt_ptr);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = (blockIdx.x * 4 + wave) * 64;
    if (q0 >= nq) return;
    const int nsplit = gridDim.y, sp = blockIdx.y;
    const int ntiles = (nt + 31) >> 5;
    const int tchunk = (ntiles + nsplit - 1) / nsplit;
    const int tile0 = sp * tchunk, tile1 = min(ntiles, tile0 + tchunk);
    const int col = lane & 31, half = lane >> 5;
    // B operand: lane (col, half) supplies query q0 + 32*qt + col, dims 32*half + s at k-step s (the k order is a
    // permutation of the dims, the same one for A and B); scaled by -2 (exact) so the accumulator holds -2 q.t
    float bq0[32], bq1[32];
    {
        const float4 *p0 = reinterpret_cast<const float4 *>(J.q + (size_t)min(q0 + col, nq - 1) * 64 + 32 * half);
        const float4 *p1 = reinterpret_cast<const float4 *>(J.q + (size_t)min(q0 + 32 + col, nq - 1) * 64 + 32 * half);
#pragma unroll
        for (int s4 = 0; s4 < 8; s4++) {
            const float4 u = p0[s4], v = p1[s4];
            bq0[4 * s4 + 0] = -2.f * u.x; bq0[4 * s4 + 1] = -2.f * u.y; bq0[4 * s4 + 2] = -2.f * u.z; bq0[4 * s4 + 3] = -2.f * u.w;
            bq1[4 * s4 + 0] = -2.f * v.x; bq1[4 * s4 + 1] = -2.f * v.y; bq1[4 * s4 + 2] = -2.f * v.z; bq1[4 * s4 + 3] = -2.f * v.w;
        }
    }
    const float baug = half == 0 ? 1.f : 0.f;
    const bool va = q0 + col < nq, vb = q0 + 32 + col < nq;
    BfmState Sa = {INFINITY, INFINITY, INFINITY, 0}, Sb = {INFINITY, INFINITY, INFINITY, 0};
    // lists are stored [list][entry][query] (query fastest): the 32 lanes of a half append to neighbouring addresses and
    // the verifier, one thread per query, reads them coalesced
    const size_t pitch = (size_t)J.capq;
    const int lst = sp * 2 + half;
    uint2 *lista = J.c_ent + (size_t)lst * BFM_CAPL * pitch + (q0 + col), *listb = lista + 32;
    const float *T = J.t;
    for (int pass = 0; pass < 2; pass++) {          // pass 0: first tile only, to seed the running second best
        const int tend = pass == 0 ? min(tile0 + 1, tile1) : tile1;
        float4 an[8];
        if (tile0 < tend) {
            const float4 *pn = reinterpret_cast<const float4 *>(T + (size_t)min(tile0 * 32 + col, nt - 1) * 64 + 32 * half);
#pragma unroll
            for (int s4 = 0; s4 < 8; s4++) an[s4] = pn[s4];
        }
        for (int tl = tile0; tl < tend; tl++) {
            float a[32];
#pragma unroll
            for (int s4 = 0; s4 < 8; s4++) { a[4 * s4] = an[s4].x; a[4 * s4 + 1] = an[s4].y; a[4 * s4 + 2] = an[s4].z; a[4 * s4 + 3] = an[s4].w; }
            if (tl + 1 < tend) {                    // prefetch the next train tile behind this tile's MFMAs
                const float4 *pn = reinterpret_cast<const float4 *>(T + (size_t)min((tl + 1) * 32 + col, nt - 1) * 64 + 32 * half);
#pragma unroll
                for (int s4 = 0; s4 < 8; s4++) an[s4] = pn[s4];
            }
            // |t|^2 of this lane's train row: half sums met across the two lane halves (a + b == b + a, so both agree)
            float part = 0.f;
#pragma unroll
            for (int s = 0; s < 32; s++) part += a[s] * a[s];
            const float tnorm = part + __shfl_xor(part, 32, 64);
            const float aaug = half == 0 ? (tl * 32 + col < nt ? tnorm : BFM_FAR) : 0.f;
            f16v acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
#pragma unroll
            for (int s = 0; s < 32; s++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bq0[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bq1[s], acc1, 0, 0, 0);
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aaug, baug, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aaug, baug, acc1, 0, 0, 0);
            if (pass == 0) {
                bfm_warm(acc0, Sa); bfm_warm(acc1, Sb);
            } else {
                const int row0 = tl * 32 + 4 * half;
                // the seed tile is already part of the running statistics: list its candidates without counting them twice
                bfm_scan(acc0, Sa, row0, lista, pitch, tl != tile0); bfm_scan(acc1, Sb, row0, listb, pitch, tl != tile0);
            }
        }
        if (pass == 0) {
            Sa.thr = va ? Sa.m2 + BFM_MARGIN : -INFINITY;      // lanes of queries beyond nq never append
            Sb.thr = vb ? Sb.m2 + BFM_MARGIN : -INFINITY;
        }
    }
    if (va) J.c_cnt[(size_t)lst * pitch + q0 + col] = Sa.cnt;
    if (vb) J.c_cnt[(size_t)lst * pitch + q0 + 32 + col] = Sb.cnt;
}

// ---------------------------------------------------------------------------------------------------
// The same filter on the bf16 matrix pipe (16x the f32-MFMA rate) with SPLIT operands: every descriptor element x is stored as
// hi = bf16(x) and lo = bf16(x - hi), and q.t is taken as hi.hi + hi.lo + lo.hi (three v_mfma_f32_32x32x16_bf16 products per 16
// dims, f32 accumulation).  What is dropped is lo.lo (<= 2^-16 |q_i t_i| per term) and the rounding of lo (2^-17 relative): for
// norms <= 1 the score is within ~3e-5 of the f32 score -- BFM_MARGIN is 30x that -- and the scores only select WHICH distances the
// verifier computes exactly.  |t|^2 rides along as one more k-slot (its own hi + lo against 1.0 on the query side).
// k_bf_split16 writes the operands once per ROI.  Queries (each wave loads its 64 once): row = [dims 0-31: hi x 32 | lo x 32 |
// dims 32-63: hi x 32 | lo x 32 | 1, 1, 0...].  Trains (every wave streams all of them) are stored in MFMA FRAGMENT order: per
// 32-train tile nine 1 KB fragments -- hi of k-steps 0..3, lo of k-steps 0..3, the |t|^2 slot -- each holding the 16 bytes of lane 0,
// lane 1, ... lane 63, so that one load instruction of a wave reads 8 consecutive cache lines instead of one line per lane (the
// row-major layout kept the texture-address path, not the matrix pipe, busy: 64 lines per load).
// ---------------------------------------------------------------------------------------------------
#define BF16_ROW 144                 // uint16 per descriptor row: 2 x (32 hi + 32 lo) + 16 for the norm slot
#define BF16_FRAGS 9                 // per train tile: 4 hi + 4 lo + norm fragments of 64 lanes x 8 uint16
typedef short s8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float x)
{
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// one thread per (descriptor row, 32-dim half); grid (ceil(cap / 128), 2 [q | t], jobs)
__global__ __launch_bounds__(256) void k_bf_split16(const MatchDev *jobs, float scale_q)
{
    const MatchDev &J = jobs[blockIdx.z];
    const bool is_t = blockIdx.y == 1;
    const int n = is_t ? *J.nt_ptr : *J.nq_ptr;
    const int row = blockIdx.x * 128 + (threadIdx.x >> 1), half = threadIdx.x & 1;
    if (row >= n) return;
    const float *src = (is_t ? J.t : J.q) + (size_t)row * 64 + 32 * half;
    const float sc = is_t ? 1.f : scale_q;                      // queries carry the factor -2 (exact in bf16)
    float part = 0.f;
    s8v hi[4], lo[4];
#pragma unroll
    for (int d = 0; d < 32; d++) {
        const float x = src[d];
        part += x * x;
        const float xs = x * sc;
        const unsigned short h = bf16_rne(xs);
        hi[d >> 3][d & 7] = (short)h;
        lo[d >> 3][d & 7] = (short)bf16_rne(xs - bf16_to_f32(h));
    }
    // norm slot (trains: |t|^2 as hi + lo; queries: 1, 1): k-slots 0 and 1 of the fifth MFMA step, lane half 0 only
    const float tot = part + __shfl_xor(part, 1, 64);
    s8v ns = {0, 0, 0, 0, 0, 0, 0, 0};
    if (half == 0) {
        unsigned short a = 0x3f80, b = 0x3f80;                    // 1.0
        if (is_t) { a = bf16_rne(tot); b = bf16_rne(tot - bf16_to_f32(a)); }
        ns[0] = (short)a; ns[1] = (short)b;
    }
    if (is_t) {
        s8v *dst = reinterpret_cast<s8v *>(J.t16) + (size_t)(row >> 5) * (BF16_FRAGS * 64) + (row & 31) + 32 * half;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) { dst[s4 * 64] = hi[s4]; dst[(4 + s4) * 64] = lo[s4]; }
        dst[8 * 64] = ns;
    } else {
        s8v *dst = reinterpret_cast<s8v *>(J.q16 + (size_t)row * BF16_ROW + 64 * half);
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) { dst[s4] = hi[s4]; dst[4 + s4] = lo[s4]; }
        if (half == 0) *reinterpret_cast<s8v *>(J.q16 + (size_t)row * BF16_ROW + 128) = ns;
    }
}

// Two launches.  PASS 0 (bounds): hi.hi products only (5 MFMAs per accumulator and tile) and a branch-free running (best, second
// best) per lane, written per (list, query).  Its scores are within BFM_HI_ERR of the split scores, so
//     thr(q) = second best over all lists + BFM_HI_ERR + BFM_MARGIN
// is an upper bound of the true second-best score plus the margin, known BEFORE the second sweep.  PASS 1 (candidates): full split
// products; a train is listed when its score is <= thr(q) -- a handful per query instead of the ~2 ln(n) records per list that a
// running threshold admits, so the append branch is almost never taken (one min3 tree + one ballot per accumulator decides) and
// the verifier has a few distances to evaluate instead of ~200.
// (measured, round 6: skipping the eight correction k-steps of tiles whose hi.hi scores all lie above the thresholds made pass 1 8 % SLOWER --
//  0.652 -> 0.703 ms on the 16-pair batch, profiles/r06_ab_bf_skip.txt: the extra minimum tree and the wave-uniform branch break the MFMA
//  chain, and the matrix pipe was only 0.54 busy to begin with; kept behind the switch)
#ifndef VFSMS_BF_SKIP
#define VFSMS_BF_SKIP 0
#endif
#ifndef BFM_GLDS
#define BFM_GLDS 1                // round 6: train tiles reach LDS by LDS-DMA (global_load_lds_dwordx4), not through registers
#endif
#define BFM_HI_ERR 1.6e-2f           // >= 2 * ((1 + 2^-8)^2 - 1) * |q||t| = 1.57e-2 (bf16 keeps 8 significand bits: RNE unit roundoff 2^-8) + the split filter's own 3e-5
#define GASM __attribute__((address_space(1)))
typedef GASM const s8v *g_cs8v;
typedef float f2v __attribute__((ext_vector_type(2)));

#ifdef VFSMS_DESC_TIMING
// debug build: per-wave cycles of the filter sweeps, [pass][0 prologue, 1 barrier wait, 2 put + fetch issue, 3 LDS reads + MFMAs + minimum tree, 4 append, 5 whole kernel, 6 waves]
__device__ unsigned long long g_bf_cycles[2][8];
extern "C" int vfsms_debug_bf_cycles(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bf_cycles), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -3; }
#define BT_DECL unsigned long long _bt[5] = {0, 0, 0, 0, 0}, _bt0 = clock64(), _btl = _bt0
#define BT_MARK(ph) do { const unsigned long long _n = clock64(); _bt[ph] += _n - _btl; _btl = _n; } while (0)
#define BT_FLUSH(pass) do { if ((threadIdx.x & 63) == 0) { for (int _q = 0; _q < 5; _q++) atomicAdd(&g_bf_cycles[pass][_q], _bt[_q]); \
        atomicAdd(&g_bf_cycles[pass][5], clock64() - _bt0); atomicAdd(&g_bf_cycles[pass][6], 1ull); } } while (0)
#else
#define BT_DECL do {} while (0)
#define BT_MARK(ph) do {} while (0)
#define BT_FLUSH(pass) do {} while (0)
#endif
template <int PASS>
__global__ __launch_bounds__(256, 3) void k_bf_mfma16_d64(const MatchDev *jobs, int qblocks, int nsplit, int njobs)
{
    // a (job, train chunk) and its 0.3 MB of split descriptors stay on one XCD (xcd_roi_map): every query block of the unit re-reads them
    unsigned unit, qb;
    xcd_roi_map(blockIdx.x, (unsigned)qblocks, (unsigned)(nsplit * njobs), unit, qb);
    const int sp = (int)(unit % (unsigned)nsplit);
    const MatchDev &J = jobs[unit / (unsigned)nsplit];
    const int nq = __builtin_amdgcn_readfirstlane(*J.nq_ptr), nt = __builtin_amdgcn_readfirstlane(*J.nt_ptr);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)qb * 256 >= nq) return;                    // (whole workgroup)
    BT_DECL;
    const int q0 = ((int)qb * 4 + wave) * 64;
    const bool live = q0 < nq;                           // a wave beyond nq still stages train tiles and keeps the barriers
    const int ntiles = (nt + 31) >> 5;
    const int tchunk = (ntiles + nsplit - 1) / nsplit;
    const int tile0 = sp * tchunk, tile1 = min(ntiles, tile0 + tchunk);
    const int col = lane & 31, half = lane >> 5;
    const s8v zero = {0, 0, 0, 0, 0, 0, 0, 0};
    // B operands (queries, resident): per 16-dim step s the lane's 8 hi and 8 lo values of dims 32*half + 8*s .. + 7
    s8v bh0[4], bl0[4], bh1[4], bl1[4], bn0, bn1;
    {
        g_cs8v p0 = (g_cs8v)(J.q16 + (size_t)min(q0 + col, nq - 1) * BF16_ROW + 64 * half);
        g_cs8v p1 = (g_cs8v)(J.q16 + (size_t)min(q0 + 32 + col, nq - 1) * BF16_ROW + 64 * half);
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            bh0[s4] = p0[s4]; bh1[s4] = p1[s4];
            if (PASS == 1) { bl0[s4] = p0[4 + s4]; bl1[s4] = p1[4 + s4]; } else { bl0[s4] = zero; bl1[s4] = zero; }
        }
        bn0 = half == 0 ? *(g_cs8v)(J.q16 + (size_t)min(q0 + col, nq - 1) * BF16_ROW + 128) : zero;
        bn1 = half == 0 ? *(g_cs8v)(J.q16 + (size_t)min(q0 + 32 + col, nq - 1) * BF16_ROW + 128) : zero;
    }
    // Train tiles go through LDS: the four waves of a workgroup sweep the SAME tiles (for different queries), so one cooperative
    // copy per tile (2.25 x 16 B per thread, straight in fragment order) replaces four sets of per-wave loads, and the operands
    // arrive by ds_read_b128 instead of waiting on L2.  The waves meet once per UNIT of BFM_UNIT tiles (they sit on four SIMDs, each
    // shared with other workgroups: every meeting waits for the slowest): two buffers of one unit each; unit u + 1 is put into the
    // buffer that unit u - 1 vacated right after the barrier (it was fetched into registers an iteration ago), unit u + 2 is fetched
    // before the MFMAs of unit u.
    constexpr int NFR = PASS == 0 ? 5 : BF16_FRAGS;      // fragments staged: hi x 4 (+ lo x 4) + norm
    constexpr int BFM_UNIT = 2;                          // train tiles per meeting (four in pass 0 -- twice the bytes in flight -- changed nothing)
    __shared__ s8v stage[2][BFM_UNIT][NFR * 64];
    g_cs8v T = (g_cs8v)J.t16;                            // fragment order: tile * 9 fragments * 64 lanes (k_bf_split16)
    const int tid = threadIdx.x; (void)tid;
    // Units are fetched BFM_SETS + 1 ahead of their MFMAs into BFM_SETS register sets (pass 0 has the registers for two: its iteration --
    // ten MFMAs per tile -- is shorter than a trip to memory, and half of its wave cycles were the wait in front of the put).
#if BFM_GLDS
    // Round 6: a tile's fragments are ONE lane-linear block in global memory (k_bf_split16 writes them in fragment order) and the same block
    // in LDS, so the copy is LDS-DMA: wave w issues global_load_lds_dwordx4 for the 1 KB chunks w and 4 + w (wave 0 also the norm
    // fragment) -- two or three instructions per tile and wave, no staging registers (24 VGPRs), no ds_write pass.  Unit u + 1 is
    // requested right behind the barrier that frees its buffer and has the whole of unit u's MFMAs to land; the barrier's vmcnt(0)
    // (hipcc drains the DMA in front of __syncthreads) is what makes it visible.  (profiles/r06_glds_probe.txt: the instruction takes
    // 4-byte-aligned global addresses and partial EXEC.)
    constexpr int NSET = 1;
    const int wv_u = __builtin_amdgcn_readfirstlane(wave);
    auto glds16 = [&](g_cs8v src, const s8v *dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    auto fetch = [&](int unit, int b) {
#pragma unroll
        for (int u = 0; u < BFM_UNIT; u++) {
            const int tl = min(tile0 + unit * BFM_UNIT + u, tile1 - 1);
            g_cs8v pn = T + (size_t)tl * (BF16_FRAGS * 64);
            glds16(pn + 64 * wv_u + lane, &stage[b][u][64 * wv_u]);
            if (PASS == 1) {
                glds16(pn + 256 + 64 * wv_u + lane, &stage[b][u][256 + 64 * wv_u]);
                if (wv_u == 0) glds16(pn + 512 + lane, &stage[b][u][PASS == 1 ? 512 : 0]);
            } else if (wv_u == 0) glds16(pn + 512 + lane, &stage[b][u][256]);
        }
    };
    const int nunits = (tile1 - tile0 + BFM_UNIT - 1) / BFM_UNIT;
    if (nunits > 0) fetch(0, 0);
#else
    constexpr int NSET = PASS == 0 ? 2 : 1;
    s8v g0[NSET][BFM_UNIT], g1[NSET][BFM_UNIT], g2[NSET][BFM_UNIT];
#pragma unroll
    for (int e = 0; e < NSET; e++)
#pragma unroll
        for (int u = 0; u < BFM_UNIT; u++) { g0[e][u] = zero; g1[e][u] = zero; g2[e][u] = zero; }
    const int nunits = (tile1 - tile0 + BFM_UNIT - 1) / BFM_UNIT;
    // (no branch around a load: beyond the chunk the last tile is fetched again and never used, every thread fetches a norm fragment lane --
    //  with a fixed number of loads per fetch the compiler waits for exactly the set it is about to put, not for everything outstanding)
    auto fetch = [&](int unit, int e) {
#pragma unroll
        for (int u = 0; u < BFM_UNIT; u++) {
            const int tl = min(tile0 + unit * BFM_UNIT + u, tile1 - 1);
            g_cs8v pn = T + (size_t)tl * (BF16_FRAGS * 64);
            g0[e][u] = pn[tid];
            if (PASS == 1) { g1[e][u] = pn[256 + tid]; g2[e][u] = pn[512 + (tid & 63)]; }
            else g1[e][u] = pn[512 + (tid & 63)];
        }
    };
    auto put = [&](int b, int e) {
#pragma unroll
        for (int u = 0; u < BFM_UNIT; u++) {
            stage[b][u][tid] = g0[e][u];
            if (PASS == 1) { stage[b][u][256 + tid] = g1[e][u]; if (tid < 64) stage[b][u][512 + tid] = g2[e][u]; }
            else if (tid < 64) stage[b][u][256 + tid] = g1[e][u];
        }
    };
    if (nunits > 0) fetch(0, 0);                     // under way while the query operands and the thresholds arrive
#endif
    // Pin the query operands as "defined here": the compiler otherwise carries their load waits into the tile loop as in-order
    // vmcnt counts, which also drain the train prefetch issued at the top of every iteration (a full L2 round trip per tile).
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
        asm volatile("" : "+v"(bh0[s4]), "+v"(bh1[s4]));
        if (PASS == 1) asm volatile("" : "+v"(bl0[s4]), "+v"(bl1[s4]));
    }
    asm volatile("" : "+v"(bn0), "+v"(bn1));
    const bool va = q0 + col < nq, vb = q0 + 32 + col < nq;
    const size_t pitch = (size_t)J.capq;
    const int lst = sp * 2 + half, nl = 2 * nsplit;
    float m1a = INFINITY, m2a = INFINITY, m1b = INFINITY, m2b = INFINITY;       // PASS 0: running best / second best
    float thra = -INFINITY, thrb = -INFINITY;                                  // PASS 1: fixed thresholds (lanes beyond nq never append)
    int cnta = 0, cntb = 0;
    if (PASS == 1) {
        GASM const f2v *B = (GASM const f2v *)J.c_m12;
        float a1 = INFINITY, a2 = INFINITY, b1 = INFINITY, b2 = INFINITY;
        for (int L0 = 0; L0 < nl; L0 += 4) {                  // four lists per round trip (the loop was one dependent load per list)
            f2v ua[4], ub[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const size_t row = (size_t)min(L0 + k, nl - 1) * pitch;
                ua[k] = B[row + min(q0 + col, nq - 1)]; ub[k] = B[row + min(q0 + 32 + col, nq - 1)];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (L0 + k >= nl) break;
                a2 = fminf(fminf(a2, ua[k].y), fmaxf(a1, ua[k].x)); a1 = fminf(a1, ua[k].x);
                b2 = fminf(fminf(b2, ub[k].y), fmaxf(b1, ub[k].x)); b1 = fminf(b1, ub[k].x);
            }
        }
        if (va) thra = a2 + (BFM_HI_ERR + BFM_MARGIN);
        if (vb) thrb = b2 + (BFM_HI_ERR + BFM_MARGIN);
    }
    // (global address space: a generic pointer makes the append a flat_store, which the compiler orders behind every LDS-DMA in flight)
    GASM unsigned long long *lista = (GASM unsigned long long *)J.c_ent + (size_t)lst * BFM_CAPL * pitch + (q0 + col), *listb = lista + 32;   // uint2 (score bits, train) as one 64-bit word
#if !BFM_GLDS
    if (nunits > 0) {
        put(0, 0);
#pragma unroll
        for (int k = 1; k <= NSET; k++) fetch(min(k, nunits - 1), k % NSET);
    }
#endif
    BT_MARK(0);
    for (int unit0 = 0; unit0 < nunits; unit0 += NSET) {
#pragma unroll
      for (int v = 0; v < NSET; v++) {                 // unrolled: the register set of unit + 1 is (v + 1) % NSET, a constant
        const int unit = unit0 + v;
        if (unit >= nunits) break;
        const int b = unit & 1;
#if BFM_GLDS
        // this wave's LDS-DMA of the unit has landed (hipcc 7.2 does NOT put this wait in front of the barrier by itself: the ISA showed
        // `s_waitcnt lgkmcnt(0); s_barrier` only) -- behind the barrier every wave's has
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        BT_MARK(1);
#if BFM_GLDS
        if (unit + 1 < nunits) fetch(unit + 1, b ^ 1);
#else
        if (unit + 1 < nunits) put(b ^ 1, (v + 1) % NSET);
        fetch(min(unit + 1 + NSET, nunits - 1), (v + 1) % NSET);
#endif
        BT_MARK(2);
        if (!live) continue;
#pragma unroll
        for (int u = 0; u < BFM_UNIT; u++) {
            const int tl = tile0 + unit * BFM_UNIT + u;
            if (tl >= tile1) break;
            const s8v *st = stage[b][u];
            f16v acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
#if VFSMS_BF_SKIP
            // Round 6 (pass 1): hi.hi + |t|^2 first -- the five k-steps of the bounds pass -- and the eight correction steps (hi.lo, lo.hi)
            // only for an accumulator in which some query could still reach its threshold: the correction moves a score by at most
            // BFM_HI_ERR, so a tile whose hi.hi scores all lie above thr + BFM_HI_ERR + BFM_MARGIN lists nothing.  Thresholds sit a hair
            // above a query's second-best score: most 32 x 32 tiles hold no such train.  (The order of the k-steps changes the rounding of
            // a listed score by ~1e-6, far inside BFM_MARGIN; what the verifier computes from the lists is exact either way.)
            if (PASS == 1) {
                s8v ah[4];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    ah[s] = st[s * 64 + lane];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh0[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh1[s], acc1, 0, 0, 0);
                }
                const s8v na = st[(NFR - 1) * 64 + lane];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(na, bn0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(na, bn1, acc1, 0, 0, 0);
                float pa = fminf(fminf(acc0[0], acc0[1]), fminf(acc0[2], acc0[3])), pb = fminf(fminf(acc1[0], acc1[1]), fminf(acc1[2], acc1[3]));
#pragma unroll
                for (int i = 4; i < 16; i += 3) {
                    pa = fminf(pa, fminf(acc0[i], fminf(acc0[i + 1], acc0[i + 2])));
                    pb = fminf(pb, fminf(acc1[i], fminf(acc1[i + 1], acc1[i + 2])));
                }
                const bool need0 = __any(pa <= thra + (BFM_HI_ERR + BFM_MARGIN)), need1 = __any(pb <= thrb + (BFM_HI_ERR + BFM_MARGIN));
                if (!need0 && !need1) continue;
                if (need0) {
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const s8v al = st[(4 + s) * 64 + lane];
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl0[s], acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0[s], acc0, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; i++) acc0[i] = BFM_FAR;
                }
                if (need1) {
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const s8v al = st[(4 + s) * 64 + lane];
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl1[s], acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1[s], acc1, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; i++) acc1[i] = BFM_FAR;
                }
            } else
#endif
            {
#pragma unroll
            for (int s = 0; s < 4; s++) {            // fragment s = hi, 4 + s = lo of the lane's dims 8s .. 8s + 7
                const s8v ah = st[s * 64 + lane];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1[s], acc1, 0, 0, 0);
                if (PASS == 1) {
                    const s8v al = st[(4 + s) * 64 + lane];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl0[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl1[s], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1[s], acc1, 0, 0, 0);
                }
            }
            const s8v na = st[(NFR - 1) * 64 + lane];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(na, bn0, acc0, 0, 0, 0);      // + |t|^2 (hi + lo in both passes)
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(na, bn1, acc1, 0, 0, 0);
            }
            if (tl * 32 + 31 >= nt) {                // train rows beyond nt never qualify
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const bool past = tl * 32 + 4 * half + (i & 3) + 8 * (i >> 2) >= nt;
                    acc0[i] = past ? BFM_FAR : acc0[i]; acc1[i] = past ? BFM_FAR : acc1[i];
                }
            }
            if (PASS == 0) {
                // A lane's 16 scores of an accumulator belong to ONE query.  Only their minimum enters the running (best, second best): the
                // second smallest of a SUBSET of the scores (one per tile and lane) is >= the true second smallest, so thr(q) stays an upper
                // bound -- exact unless a query's two nearest trains share a tile and a lane half (15 / nt of the queries; pass 1 then lists a
                // few more candidates for them) -- and the sweep costs 11 VALU operations per accumulator instead of 48 (it was VALU-bound).
                float ta = fminf(fminf(acc0[0], acc0[1]), fminf(acc0[2], acc0[3])), tb = fminf(fminf(acc1[0], acc1[1]), fminf(acc1[2], acc1[3]));
#pragma unroll
                for (int i = 4; i < 16; i += 3) {
                    ta = fminf(ta, fminf(acc0[i], fminf(acc0[i + 1], acc0[i + 2])));
                    tb = fminf(tb, fminf(acc1[i], fminf(acc1[i + 1], acc1[i + 2])));
                }
                m2a = fminf(m2a, fmaxf(m1a, ta)); m1a = fminf(m1a, ta);
                m2b = fminf(m2b, fmaxf(m1b, tb)); m1b = fminf(m1b, tb);
            } else {
                const int row0 = tl * 32 + 4 * half;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const f16v &acc = h == 0 ? acc0 : acc1;
                    const float thr = h == 0 ? thra : thrb;
                    // minima of the value groups {0..3}, {4..6}, {7..9}, {10..12}, {13..15}: the append walks only the groups that hold a hit
                    // (one or two values of the 16, as a rule), in ascending order -- the lists are the ones a full walk writes
                    float gm[5];
                    gm[0] = fminf(fminf(acc[0], acc[1]), fminf(acc[2], acc[3]));
#pragma unroll
                    for (int g = 1; g < 5; g++) gm[g] = fminf(acc[3 * g + 1], fminf(acc[3 * g + 2], acc[3 * g + 3]));
                    const float mn = fminf(fminf(gm[0], fminf(gm[1], gm[2])), fminf(gm[3], gm[4]));
                    const bool some = __any(mn <= thr);
                    BT_MARK(3);
                    if (some) {
                        GASM unsigned long long *list = h == 0 ? lista : listb;
                        int cnt = h == 0 ? cnta : cntb;
#pragma unroll
                        for (int g = 0; g < 5; g++) {
                            if (!__any(gm[g] <= thr)) continue;
#pragma unroll
                            for (int i = (g == 0 ? 0 : 3 * g + 1); i < (g == 0 ? 4 : 3 * g + 4); i++) {
                                const float v = acc[i];
                                if (v <= thr && v < 1e29f) {
                                    if (cnt < BFM_CAPL) list[cnt * pitch] = (unsigned long long)__float_as_uint(v) | ((unsigned long long)(unsigned)(row0 + (i & 3) + 8 * (i >> 2)) << 32);
                                    cnt++;
                                }
                            }
                        }
                        if (h == 0) cnta = cnt; else cntb = cnt;
                        BT_MARK(4);
                    }
                }
            }
            if (PASS == 0) BT_MARK(3);
        }
      }
    }
    BT_FLUSH(PASS);
    if (PASS == 0) {
        GASM f2v *B = (GASM f2v *)J.c_m12;
        if (va) B[(size_t)lst * pitch + q0 + col] = f2v{m1a, m2a};
        if (vb) B[(size_t)lst * pitch + q0 + 32 + col] = f2v{m1b, m2b};
    } else {
        if (va) J.c_cnt[(size_t)lst * pitch + q0 + col] = cnta;
        if (vb) J.c_cnt[(size_t)lst * pitch + q0 + 32 + col] = cntb;
    }
}

// One THREAD per query (no cross-lane traffic; list reads are coalesced across the queries of a wave).  Sweep 1 takes
// the second smallest score S2 over the query's lists (the lists hold the true 2-NN, so S2 bounds their scores up to the
// MFMA rounding); sweep 2 parks the entries within BFM_MARGIN of S2 -- typically two or three -- in a per-thread LDS
// column; the exact distances are then evaluated hit by hit, all lanes in step, with the arithmetic of k_bf_l2_d64 and
// the semantics of knn_update over ascending train indices (best = lowest index among the smallest sqrt-domain
// distances; second = next smallest value), which is order-independent in this form.
#define BFV_HITS 12
#ifdef VFSMS_DESC_TIMING
__device__ unsigned g_bfv_stats[4];        // debug build: [0] queries, [1] queries with an overflowed list, [2] list entries read, [3] exact evaluations
extern "C" int vfsms_debug_bfv_stats(unsigned *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bfv_stats), sizeof(unsigned) * 4) == hipSuccess ? 0 : -3; }
#define BFV_STAT(i, v) atomicAdd(&g_bfv_stats[i], (unsigned)(v))
#else
#define BFV_STAT(i, v) do {} while (0)
#endif
__device__ __forceinline__ void bfv_exact(const float *__restrict__ Qr, const float *__restrict__ T, int idx, float &B1, float &B2, int &I1)
{
    const float4 *tr = reinterpret_cast<const float4 *>(T + (size_t)idx * 64);
    const float4 *qr = reinterpret_cast<const float4 *>(Qr);
    float acc = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < 16; d4++) {
        const float4 tv = tr[d4], qv = qr[d4];
        const float e0 = qv.x - tv.x, e1 = qv.y - tv.y, e2 = qv.z - tv.z, e3 = qv.w - tv.w;
        acc += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    }
    const float d = sqrtf(acc);
    if (d < B1 || (d == B1 && idx < I1)) { B2 = B1; B1 = d; I1 = idx; }
    else B2 = fminf(B2, d);
}

__global__ __launch_bounds__(256) void k_bf_verify_d64(const MatchDev *jobs, int cns)
{
    const MatchDev &J = jobs[blockIdx.y];
    const int nq = __builtin_amdgcn_readfirstlane(*J.nq_ptr), nt = __builtin_amdgcn_readfirstlane(*J.nt_ptr);
    if ((int)(blockIdx.x * 256) >= nq) return;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const bool live = q < nq;
    const int nl = 2 * cns;
    const size_t pitch = (size_t)J.capq;
    __shared__ int hits[BFV_HITS][256];
    float s1 = INFINITY, s2 = INFINITY; bool overflow = false;
    if (live)
        for (int L = 0; L < nl; L++) {
            const int c = J.c_cnt[(size_t)L * pitch + q];
            overflow = overflow || c > BFM_CAPL;
            const uint2 *e = J.c_ent + (size_t)L * BFM_CAPL * pitch + q;
            for (int k = 0; k < min(c, BFM_CAPL); k++) {
                const float v = __uint_as_float(e[k * pitch].x);
                s2 = fminf(s2, fmaxf(s1, v));
                s1 = fminf(s1, v);
            }
        }
    const float cut = s2 + BFM_MARGIN;
    if (live) { BFV_STAT(0, 1); BFV_STAT(1, overflow ? 1 : 0); }
    const float *Qr = J.q + (size_t)min(q, nq - 1) * 64;
    float B1 = INFINITY, B2 = INFINITY; int I1 = -1;
    int nh = 0;
    if (live && !overflow)
        for (int L = 0; L < nl; L++) {
            const int c = min(J.c_cnt[(size_t)L * pitch + q], BFM_CAPL);
            const uint2 *e = J.c_ent + (size_t)L * BFM_CAPL * pitch + q;
            for (int k = 0; k < c; k++) {
                const uint2 ev = e[k * pitch];
                if (__uint_as_float(ev.x) <= cut) {
                    if (nh < BFV_HITS) hits[nh++][threadIdx.x] = (int)ev.y;
                    else bfv_exact(Qr, J.t, (int)ev.y, B1, B2, I1);      // more near-ties than slots: evaluate in place
                }
            }
        }
    if (live) BFV_STAT(3, nh);
    int maxh = nh;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxh = max(maxh, __shfl_xor(maxh, d, 64));
    for (int k = 0; k < maxh; k++)
        if (k < nh) bfv_exact(Qr, J.t, hits[k][threadIdx.x], B1, B2, I1);
    // A list overflowed (a query with dozens of trains inside the margin, e.g. an all-zero descriptor): exhaustive exact scan of
    // all trains for that query, shared by the 64 lanes of its wave and merged with the same first-index / second-value rule.
    {
        const int lane = threadIdx.x & 63;
        unsigned long long om = __ballot(live && overflow);
        while (om) {
            const int src = __ffsll((long long)om) - 1;
            om &= om - 1;
            const int qo = __shfl(q, src, 64);
            const float *Qo = J.q + (size_t)qo * 64;
            float b1 = INFINITY, b2 = INFINITY; int i1 = -1;
            for (int j = lane; j < nt; j += 64) bfv_exact(Qo, J.t, j, b1, b2, i1);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const float ob1 = __shfl_xor(b1, d, 64), ob2 = __shfl_xor(b2, d, 64);
                const int oi1 = __shfl_xor(i1, d, 64);
                if (ob1 < b1 || (ob1 == b1 && oi1 >= 0 && (i1 < 0 || oi1 < i1))) { b2 = fminf(b1, ob2); b1 = ob1; i1 = oi1; }
                else b2 = fminf(b2, ob1);
            }
            if (lane == src) { B1 = b1; B2 = b2; I1 = i1; }
        }
    }
    if (live) { J.p_d1[q] = B1; J.p_d2[q] = B2; J.p_i1[q] = I1; }
}

// merge the per-split 2-NN lists, apply the ratio test (Python double arithmetic on float32 distances,
// ImageUtility.py:294) and compute the vote of ImageUtility.py:153-161 for surviving matches.
__global__ __launch_bounds__(256) void k_merge_ratio(const MatchDev *jobs, double ratio, int do_ratio)
{
    const MatchDev &J = jobs[blockIdx.y];
    // device-side counts are wave-uniform: pin them to SGPRs so loop bounds and train addresses stay scalar
    const int nq = __builtin_amdgcn_readfirstlane(*J.nq_ptr), nt = __builtin_amdgcn_readfirstlane(*J.nt_ptr);
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    float B1 = INFINITY, B2 = INFINITY; int I1 = -1;
    for (int s = 0; s < J.nsplit; s++) {
        const size_t o = (size_t)s * J.capq + q;
        float d = J.p_d1[o]; int i = J.p_i1[o];
        if (i >= 0) {
            if (d < B1) { B2 = B1; B1 = d; I1 = i; } else if (d < B2) { B2 = d; }
            float e = J.p_d2[o];
            if (e < B1) { B2 = B1; B1 = e; } else if (e < B2) { B2 = e; }
        }
    }
    J.d1[q] = B1; J.d2[q] = B2; J.i1[q] = I1;
    if (!do_ratio) return;
    int ok = (I1 >= 0) && (nt >= 2) && ((double)B1 < (double)B2 * ratio);
    int vote = 0;
    if (ok && J.kq) {
        // ptA = (kpsA[q][1], kpsA[q][0]); dx = int(ptA[0]-ptB[0]) (float32 subtract, trunc toward zero)
        float ay = J.kq[2 * q + 1], ax = J.kq[2 * q];
        float by = J.kt[2 * I1 + 1], bx = J.kt[2 * I1];
        int dx = (int)(ay - by), dy = (int)(ax - bx);
        vote = !(dx == 0 && dy == 0);
        J.votes[2 * (size_t)(J.capq + q)] = dx;             // staging area: second half of the votes buffer
        J.votes[2 * (size_t)(J.capq + q) + 1] = dy;
    }
    J.match_flag[q] = ok | (vote << 1);
}

__device__ __forceinline__ int wave_incl_scan_i(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// order-preserving compaction of matches and of non-(0,0) votes: one 1024-thread workgroup per job
__global__ __launch_bounds__(1024) void k_match_scan(const MatchDev *jobs)
{
    const MatchDev &J = jobs[blockIdx.x];
    const int nq = *J.nq_ptr;
    __shared__ int wsum_m[16], wsum_v[16];
    __shared__ int carry_m, carry_v;
    if (threadIdx.x == 0) { carry_m = 0; carry_v = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int base = 0; base < nq; base += 1024) {
        const int q = base + threadIdx.x;
        const int f = (q < nq) ? J.match_flag[q] : 0;
        const int fm = f & 1, fv = (f >> 1) & 1;
        const int im = wave_incl_scan_i(fm), iv = wave_incl_scan_i(fv);
        if (lane == 63) { wsum_m[wid] = im; wsum_v[wid] = iv; }
        __syncthreads();
        int om = carry_m, ov = carry_v;
        for (int k = 0; k < wid; k++) { om += wsum_m[k]; ov += wsum_v[k]; }
        if (fm && !J.pairs_given) {
            const int pos = om + im - 1;
            J.pairs[2 * (size_t)pos] = J.i1[q];              // (trainIdx, queryIdx)
            J.pairs[2 * (size_t)pos + 1] = q;
        }
        if (fv) {
            const int pos = ov + iv - 1;
            J.votes[2 * (size_t)pos] = J.votes[2 * (size_t)(J.capq + q)];
            J.votes[2 * (size_t)pos + 1] = J.votes[2 * (size_t)(J.capq + q) + 1];
        }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_m = om + im; carry_v = ov + iv; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { J.mcount[0] = carry_m; J.mcount[1] = carry_v; J.mcount[2] = 0; J.mcount[3] = 0; }
}

// votes from an explicit (trainIdx, queryIdx) list (per-operator entry point vfsms_mode_offset)
__global__ __launch_bounds__(256) void k_votes_from_pairs(const MatchDev *jobs, int m)
{
    const MatchDev &J = jobs[blockIdx.y];
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    const int tr = J.pairs[2 * k], q = J.pairs[2 * k + 1];
    float ay = J.kq[2 * q + 1], ax = J.kq[2 * q];
    float by = J.kt[2 * tr + 1], bx = J.kt[2 * tr];
    int dx = (int)(ay - by), dy = (int)(ax - bx);
    J.votes[2 * (size_t)(J.capq + k)] = dx;
    J.votes[2 * (size_t)(J.capq + k) + 1] = dy;
    J.match_flag[k] = 1 | ((!(dx == 0 && dy == 0)) << 1);
}

// Compaction + mode vote + result record of one job in ONE 1024-thread workgroup (round 2: k_match_scan + k_mode_count + k_mode_final): after the order-preserving compaction every vote (dx, dy) is inserted into an LDS hash table
// (linear probing on the packed 32-bit key; per slot a count and the smallest vote index), then the slots bid (count, -first index):
// the most frequent tuple, ties to the first inserted -- Method.getOffsetByMode's dict order + stable sort (ImageUtility.py:165-168) --
// in O(M) instead of the O(M^2) equality count (100 us per launch on SURF batches, 230 us on ORB's 5000 unconditional votes).
#define MODE_SLOTS 8192
#define MODE_HASH_MAX 5600         // votes per pass of the table (ORB votes 5000 times: one pass); more are dealt to several passes
// three uint32[8192] tables = 96 KB of static LDS: this kernel needs gfx950's 160 KB per CU (the library is gfx950-only: README, Makefile);
// a 64 KB-LDS target would have to shrink MODE_SLOTS and run more passes (MODE_HASH_MAX)
static_assert(3 * MODE_SLOTS * sizeof(uint32_t) <= 160 * 1024 - 16 * 1024, "k_scan_mode's hash tables are sized for the 160 KB LDS of gfx950");
__global__ __launch_bounds__(1024) void k_scan_mode(const MatchDev *jobs, int offset_evaluate)
{
    const MatchDev &J = jobs[blockIdx.x];
    const int nq = *J.nq_ptr;
    __shared__ int wsum_m[16], wsum_v[16];
    __shared__ int carry_m, carry_v;
    __shared__ uint32_t hkey[MODE_SLOTS], hcnt[MODE_SLOTS], hfirst[MODE_SLOTS];
    __shared__ unsigned long long best;
    if (threadIdx.x == 0) { carry_m = 0; carry_v = 0; best = 0ull; }
    for (int s = threadIdx.x; s < MODE_SLOTS; s += 1024) { hkey[s] = 0u; hcnt[s] = 0u; hfirst[s] = 0xFFFFFFFFu; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int base = 0; base < nq; base += 1024) {
        const int q = base + threadIdx.x;
        const int f = (q < nq) ? J.match_flag[q] : 0;
        const int fm = f & 1, fv = (f >> 1) & 1;
        const int im = wave_incl_scan_i(fm), iv = wave_incl_scan_i(fv);
        if (lane == 63) { wsum_m[wid] = im; wsum_v[wid] = iv; }
        __syncthreads();
        int om = carry_m, ov = carry_v;
        for (int k = 0; k < wid; k++) { om += wsum_m[k]; ov += wsum_v[k]; }
        if (fm && !J.pairs_given) {
            const int pos = om + im - 1;
            J.pairs[2 * (size_t)pos] = J.i1[q];              // (trainIdx, queryIdx)
            J.pairs[2 * (size_t)pos + 1] = q;
        }
        if (fv) {
            const int pos = ov + iv - 1;
            J.votes[2 * (size_t)pos] = J.votes[2 * (size_t)(J.capq + q)];
            J.votes[2 * (size_t)pos + 1] = J.votes[2 * (size_t)(J.capq + q) + 1];
        }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_m = om + im; carry_v = ov + iv; }
        __syncthreads();
    }
    const int nm = carry_m, nv = carry_v;
    if (threadIdx.x == 0) { J.mcount[0] = nm; J.mcount[1] = nv; J.mcount[2] = 0; J.mcount[3] = 0; }
    __threadfence_block();
    __syncthreads();
    // insert: key = (dx + 32768) << 16 | (dy + 32768); offsets beyond +-32767 px cannot occur (tiles are <= 8192 px).
    // More votes than one table takes (configs[4]: 37 k keypoints per 819 x 4096 strip leave 6-9 k matches; round 4 sent those
    // through an O(M^2) equality count, 25 ms per launch): the votes are dealt to P passes by a second hash of the key -- equal tuples meet in
    // one pass, every pass fills a cleared table and bids (count, -first index) into `best`, so the winner is still the most frequent tuple,
    // ties to the first inserted (ImageUtility.py:165-168).
    const int P = (nv + MODE_HASH_MAX - 1) / MODE_HASH_MAX;
    for (int pass = 0; pass < P; pass++) {
        if (pass > 0) {
            __syncthreads();
            for (int s = threadIdx.x; s < MODE_SLOTS; s += 1024) { hkey[s] = 0u; hcnt[s] = 0u; hfirst[s] = 0xFFFFFFFFu; }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < nv; i += 1024) {
            const int dx = J.votes[2 * (size_t)i], dy = J.votes[2 * (size_t)i + 1];
            const uint32_t key = ((uint32_t)(dx + 32768) << 16) | ((uint32_t)(dy + 32768) & 0xffffu);
            if (P > 1 && (int)(((key ^ (key >> 15)) * 0x9E3779B1u >> 8) % (uint32_t)P) != pass) continue;
            uint32_t h = (key * 2654435761u) >> 19;             // 13 bits
            for (;;) {
                const uint32_t old = atomicCAS(&hkey[h], 0u, key);
                if (old == 0u || old == key) { atomicAdd(&hcnt[h], 1u); atomicMin(&hfirst[h], (uint32_t)i); break; }
                h = (h + 1) & (MODE_SLOTS - 1);
            }
        }
        __syncthreads();
        unsigned long long mine = 0ull;
        for (int s = threadIdx.x; s < MODE_SLOTS; s += 1024)
            if (hcnt[s]) {
                const unsigned long long bid = ((unsigned long long)hcnt[s] << 32) | (unsigned long long)(0xFFFFFFFFu - hfirst[s]);
                mine = bid > mine ? bid : mine;
            }
        for (int d = 32; d > 0; d >>= 1) { const unsigned long long o = __shfl_down(mine, d, 64); mine = o > mine ? o : mine; }
        if (lane == 0 && mine) atomicMax(&best, mine);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int status = 0, dx = 0, dy = 0, votes = 0;
        if (nm > 0) {
            if (nv == 0) { votes = 1; }                      // dxList.append(0); dyList.append(0)
            else {
                votes = (int)(best >> 32);
                const unsigned idx = 0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull);
                dx = J.votes[2 * (size_t)idx]; dy = J.votes[2 * (size_t)idx + 1];
            }
            status = votes >= offset_evaluate;
        }
        *reinterpret_cast<unsigned long long *>(J.mcount + 2) = best;
        J.result[0] = status; J.result[1] = dx; J.result[2] = dy; J.result[3] = votes;
        J.result[4] = *J.nq_ptr; J.result[5] = *J.nt_ptr; J.result[6] = nm; J.result[7] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------
// Hamming 1-NN: lanes own queries (32 bytes = 8 dwords in VGPRs), trains stream through the scalar path
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bf_hamming32(const uint32_t *__restrict__ q, int nq,
                                                      const uint32_t *__restrict__ t, int nt,
                                                      int *best_idx, int *best_dist)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= nq) return;
    const uint32_t *pq = q + (size_t)min(qi, nq - 1) * 8;
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = pq[k];
    int best = 0x7fffffff, bi = -1;
    for (int j = 0; j < nt; j++) {
        const uint32_t *__restrict__ tr = t + (size_t)j * 8;
        int d = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) d += __popc(v[k] ^ tr[k]);
        if (d < best) { best = d; bi = j; }
    }
    if (qi < nq) { best_idx[qi] = bi; best_dist[qi] = best; }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

size_t match_bytes(int capq, int nsplit)
{
    size_t b = 3 * al(sizeof(float) * (size_t)capq * nsplit);
    b += 3 * al(sizeof(float) * capq) + 2 * al(sizeof(int) * capq);
    b += al(sizeof(int32_t) * 2 * capq) + al(sizeof(int32_t) * 4 * capq) + al(64) + al(64);
    return b + 4096;
}

int match_carve(vfsms_ctx *ctx, MatchDev *m, int capq, int dim, int nsplit)
{
    m->capq = capq; m->dim = dim; m->nsplit = nsplit;
    m->p_d1 = (float *)ctx_arena_alloc(ctx, sizeof(float) * (size_t)capq * nsplit);
    m->p_d2 = (float *)ctx_arena_alloc(ctx, sizeof(float) * (size_t)capq * nsplit);
    m->p_i1 = (int *)ctx_arena_alloc(ctx, sizeof(int) * (size_t)capq * nsplit);
    m->d1 = (float *)ctx_arena_alloc(ctx, sizeof(float) * capq);
    m->d2 = (float *)ctx_arena_alloc(ctx, sizeof(float) * capq);
    m->i1 = (int *)ctx_arena_alloc(ctx, sizeof(int) * capq);
    m->match_flag = (int *)ctx_arena_alloc(ctx, sizeof(int) * capq);
    m->match_pos = (int *)ctx_arena_alloc(ctx, sizeof(int) * capq);
    m->pairs = (int32_t *)ctx_arena_alloc(ctx, sizeof(int32_t) * 2 * capq);
    m->votes = (int32_t *)ctx_arena_alloc(ctx, sizeof(int32_t) * 4 * capq);
    m->mcount = (int *)ctx_arena_alloc(ctx, 64);
    m->result = (int32_t *)ctx_arena_alloc(ctx, 64);
    if (!m->result) { vfsms_set_error("arena exhausted while carving a match job"); return VFSMS_ERR_CAPACITY; }
    return VFSMS_OK;
}

// largest squared row norm of an n x 64 array (one wave per row): decides whether host-supplied descriptors qualify
// for the MFMA-filtered search, whose margin is absolute and assumes norms <= 1
__global__ __launch_bounds__(256) void k_max_norm2_d64(const float *a, int n, unsigned *out)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float v = a[(size_t)row * 64 + lane];
    float s = v * v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) atomicMax(out, __float_as_uint(s == s ? s : INFINITY));   // non-negative floats order like their bits; NaN -> inf
}

int launch_max_norm2_d64(vfsms_ctx *ctx, const float *a, int n, unsigned *d_out)
{
    if (n > 0) hipLaunchKernelGGL(k_max_norm2_d64, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, a, n, d_out);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

size_t match_filter_bytes(int capq, int capt, int cns)
{
    return al(sizeof(uint2) * (size_t)capq * cns * 2 * BFM_CAPL) + al(sizeof(int) * (size_t)capq * cns * 2) +
           al(sizeof(unsigned short) * BF16_ROW * (size_t)capq) + al(sizeof(unsigned short) * BF16_ROW * ((size_t)capt + 32)) +
           al(sizeof(float2) * (size_t)capq * cns * 2) + 1024;
}

int match_filter_carve(vfsms_ctx *ctx, MatchDev *m, int capq, int capt, int cns)
{
    m->c_ent = (uint2 *)ctx_arena_alloc(ctx, sizeof(uint2) * (size_t)capq * cns * 2 * BFM_CAPL);
    m->c_cnt = (int *)ctx_arena_alloc(ctx, sizeof(int) * (size_t)capq * cns * 2);
    m->q16 = (unsigned short *)ctx_arena_alloc(ctx, sizeof(unsigned short) * BF16_ROW * (size_t)capq);
    m->t16 = (unsigned short *)ctx_arena_alloc(ctx, sizeof(unsigned short) * BF16_ROW * ((size_t)capt + 32));   // whole 32-train tiles
    m->c_m12 = (float2 *)ctx_arena_alloc(ctx, sizeof(float2) * (size_t)capq * cns * 2);
    if (!m->c_cnt || !m->t16 || !m->c_m12) { vfsms_set_error("arena exhausted while carving a match filter"); return VFSMS_ERR_CAPACITY; }
    return VFSMS_OK;
}

static bool bf_f32_filter()
{
    static const bool v = getenv("VFSMS_BF_F32FILTER") && atoi(getenv("VFSMS_BF_F32FILTER")) != 0;
    return v;
}

// MFMA-filtered exact 2-NN for 64-d descriptors of norm <= 1 (jobs carved with nsplit == 1 plus match_filter_carve).
// Default: split-bf16 filter on the bf16 matrix pipe; VFSMS_BF_F32FILTER=1 keeps round 1's f32-MFMA filter.
int launch_bf_l2_filtered(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, int capt, int cns)
{
    if (njobs <= 0 || capq <= 0) return VFSMS_OK;
    {
        ProfScope ps(ctx, "bf_mfma");
        if (bf_f32_filter())
            hipLaunchKernelGGL(k_bf_mfma_d64, dim3((capq + 255) / 256, cns, njobs), dim3(256), 0, ctx->stream, d_jobs);
        else {
            hipLaunchKernelGGL(k_bf_split16, dim3((std::max(capq, capt) + 127) / 128, 2, njobs), dim3(256), 0, ctx->stream, d_jobs, -2.f);
            const int qblocks = (capq + 255) / 256;
            hipLaunchKernelGGL(k_bf_mfma16_d64<0>, dim3((unsigned)qblocks * cns * njobs), dim3(256), 0, ctx->stream, d_jobs, qblocks, cns, njobs);
            hipLaunchKernelGGL(k_bf_mfma16_d64<1>, dim3((unsigned)qblocks * cns * njobs), dim3(256), 0, ctx->stream, d_jobs, qblocks, cns, njobs);
        }
    }
    {
        ProfScope ps(ctx, "bf_verify");
        hipLaunchKernelGGL(k_bf_verify_d64, dim3((capq + 255) / 256, njobs), dim3(256), 0, ctx->stream, d_jobs, cns);
    }
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

int launch_bf_l2(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, int nsplit, int dim)
{
    if (njobs <= 0 || capq <= 0) return VFSMS_OK;
    ProfScope ps(ctx, "bf_l2");
    if (dim == 64) {
        hipLaunchKernelGGL(k_bf_l2_d64, dim3((capq + 511) / 512, nsplit, njobs), dim3(256), 0, ctx->stream, d_jobs);
    } else if (dim == 128) {
        hipLaunchKernelGGL(k_bf_l2_gen<128>, dim3((capq + 255) / 256, nsplit, njobs), dim3(256), 0, ctx->stream, d_jobs);
    } else if (dim == 32) {
        hipLaunchKernelGGL(k_bf_l2_gen<32>, dim3((capq + 255) / 256, nsplit, njobs), dim3(256), 0, ctx->stream, d_jobs);
    } else {
        vfsms_set_error("bf_l2: descriptor dim %d unsupported (64, 128, 32)", dim);
        return VFSMS_ERR_UNSUPPORTED;
    }
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

// compaction + mode vote + result record: one workgroup per job
static int launch_vote_tail(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, int offset_evaluate)
{
    (void)capq;
    hipLaunchKernelGGL(k_scan_mode, dim3(njobs), dim3(1024), 0, ctx->stream, d_jobs, offset_evaluate);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

int launch_ratio_mode(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, double ratio, int offset_evaluate)
{
    if (njobs <= 0) return VFSMS_OK;
    ProfScope ps(ctx, "vote");
    hipLaunchKernelGGL(k_merge_ratio, dim3((capq + 255) / 256, njobs), dim3(256), 0, ctx->stream, d_jobs, ratio, 1);
    return launch_vote_tail(ctx, d_jobs, njobs, capq, offset_evaluate);
}

int launch_ratio_only(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, double ratio)
{
    hipLaunchKernelGGL(k_merge_ratio, dim3((capq + 255) / 256, njobs), dim3(256), 0, ctx->stream, d_jobs, ratio, 1);
    hipLaunchKernelGGL(k_match_scan, dim3(njobs), dim3(1024), 0, ctx->stream, d_jobs);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

int launch_merge_only(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq)
{
    hipLaunchKernelGGL(k_merge_ratio, dim3((capq + 255) / 256, njobs), dim3(256), 0, ctx->stream, d_jobs, 0.0, 0);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

// flags / votes already written per query (Hamming matcher): compaction + mode vote
int launch_scan_mode(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, int offset_evaluate)
{
    if (njobs <= 0) return VFSMS_OK;
    ProfScope ps(ctx, "vote");
    return launch_vote_tail(ctx, d_jobs, njobs, capq, offset_evaluate);
}

int launch_mode_only(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capm, int offset_evaluate)
{
    if (njobs <= 0) return VFSMS_OK;
    if (capm > 0)
        hipLaunchKernelGGL(k_votes_from_pairs, dim3((capm + 255) / 256, njobs), dim3(256), 0, ctx->stream, d_jobs, capm);
    return launch_vote_tail(ctx, d_jobs, njobs, capm, offset_evaluate);
}

int launch_bf_hamming(vfsms_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes,
                      int *best_idx, int *best_dist)
{
    if (nbytes != 32) { vfsms_set_error("bf_hamming: only 32-byte descriptors supported"); return VFSMS_ERR_UNSUPPORTED; }
    if (nq <= 0) return VFSMS_OK;
    hipLaunchKernelGGL(k_bf_hamming32, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream,
                       (const uint32_t *)q, nq, (const uint32_t *)t, nt, best_idx, best_dist);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}

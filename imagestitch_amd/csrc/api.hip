// api.hip -- context management and the extern "C" entry points declared in include/vfsms.h.
#include "common.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>

int fuse_i64_device(vfsms_ctx *ctx, const long long *dA, const long long *dB, int r, int c, int ch, int dx, int dy,
                    uint8_t *d_out, int32_t *info, int method = 0);

// ---- errors -------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void vfsms_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" int vfsms_last_error(char *buf, int buflen)
{
    if (!buf || buflen <= 0) return VFSMS_ERR_BAD_ARG;
    snprintf(buf, (size_t)buflen, "%s", g_err);
    return VFSMS_OK;
}
extern "C" int vfsms_version(void) { return 100; }
extern "C" int vfsms_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- arena ----------------------------------------------------------------------------------------------
int ctx_arena_reserve(vfsms_ctx *ctx, size_t bytes)
{
    bytes += 1 << 20;
    if (bytes > ctx->arena_size) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->arena) HIP_TRY(hipFree(ctx->arena));
        ctx->arena = nullptr; ctx->arena_size = 0;
        size_t want = bytes + bytes / 4;
        HIP_TRY(hipMalloc((void **)&ctx->arena, want));
        ctx->arena_size = want;
    }
    ctx->arena_off = 0;
    return VFSMS_OK;
}
void *ctx_arena_alloc(vfsms_ctx *ctx, size_t bytes, size_t align)
{
    size_t off = (ctx->arena_off + align - 1) & ~(align - 1);
    if (off + bytes > ctx->arena_size) return nullptr;
    ctx->arena_off = off + bytes;
    return ctx->arena + off;
}

// ---- per-stage profiling with HIP events on the context stream --------------------------------------------
int prof_begin(vfsms_ctx *ctx, const char *name)
{
    if (!ctx->prof_on) return -1;
    int id = -1;
    // stages enqueued on the second compute stream are booked under their own name ("bf_mfma@s2"): they run beside the first stream's stages,
    // so the per-stage times of a step no longer add up to its wall clock -- the reader sees which ones overlapped
    std::string key(name);
    if (ctx->stream2 && ctx->stream == ctx->stream2) key += "@s2";
    for (size_t k = 0; k < ctx->prof_names.size(); k++) if (ctx->prof_names[k] == key) { id = (int)k; break; }
    if (id < 0) { id = (int)ctx->prof_names.size(); ctx->prof_names.push_back(key); ctx->prof_ms.push_back(0.0); ctx->prof_calls.push_back(0); }
    ProfRec r; r.id = id;
    for (hipEvent_t *e : {&r.a, &r.b}) {
        if (!ctx->prof_pool.empty()) { *e = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); }
        else if (hipEventCreate(e) != hipSuccess) return -1;
    }
    if (hipEventRecord(r.a, ctx->stream) != hipSuccess) return -1;
    ctx->prof_recs.push_back(r);
    return (int)ctx->prof_recs.size() - 1;
}
void prof_end(vfsms_ctx *ctx, int rec)
{
    if (rec < 0 || rec >= (int)ctx->prof_recs.size()) return;
    (void)hipEventRecord(ctx->prof_recs[rec].b, ctx->stream);
}
static int prof_collect(vfsms_ctx *ctx)
{
    if (ctx->prof_recs.empty()) return VFSMS_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto &r : ctx->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { ctx->prof_ms[r.id] += ms; ctx->prof_calls[r.id] += 1; }
        ctx->prof_pool.push_back(r.a); ctx->prof_pool.push_back(r.b);
    }
    ctx->prof_recs.clear();
    return VFSMS_OK;
}
extern "C" int vfsms_profile_enable(vfsms_ctx *ctx, int on)
{
    if (!ctx) return VFSMS_ERR_BAD_ARG;
    TRY(prof_collect(ctx));
    ctx->prof_on = on != 0;
    return VFSMS_OK;
}
extern "C" int vfsms_profile_read(vfsms_ctx *ctx, char *names, int names_len, double *ms, int64_t *calls, int cap, int *n_out, int reset)
{
    if (!ctx || !n_out) return VFSMS_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    TRY(prof_collect(ctx));
    const int n = (int)ctx->prof_names.size();
    *n_out = n;
    std::string joined;
    for (int k = 0; k < n; k++) {
        if (k) joined += ",";
        joined += ctx->prof_names[k];
        if (k < cap) { if (ms) ms[k] = ctx->prof_ms[k]; if (calls) calls[k] = ctx->prof_calls[k]; }
    }
    if (names && names_len > 0) snprintf(names, (size_t)names_len, "%s", joined.c_str());
    if (reset) for (int k = 0; k < n; k++) { ctx->prof_ms[k] = 0; ctx->prof_calls[k] = 0; }
    return VFSMS_OK;
}

// ---- SURF tables (resizeHaarPattern for every layer; SURFInvoker constructor tables) -------------------
static void gaussian_kernel_f32(int n, double sigma, float *cf)   // cv::getGaussianKernel(n, sigma, CV_32F)
{
    const double scale2X = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        cf[i] = (float)exp(scale2X * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) cf[i] = (float)(cf[i] * sum);
}

int ctx_prepare_surf(vfsms_ctx *ctx, const vfsms_surf_params *p)
{
    if (!p || p->n_octaves < 1 || p->n_octave_layers < 1 || (p->n_octave_layers + 2) * p->n_octaves > VFSMS_MAX_LAYERS ||
        p->hessian_threshold < 0 || p->n_octaves > 8) {
        vfsms_set_error("bad SURF parameters");
        return VFSMS_ERR_BAD_ARG;
    }
    // largest keypoint size = largest middle-layer size + its scale step -> descriptor window side
    {
        const int lpo = p->n_octave_layers + 2;
        const int top = (9 + 6 * (lpo - 1)) << (p->n_octaves - 1);
        const float s = (float)top * 1.2f / 9.0f;
        if ((int)(21 * s) > VFSMS_MAX_WIN) {
            vfsms_set_error("SURF parameters give descriptor windows > %d px (unsupported)", VFSMS_MAX_WIN);
            return VFSMS_ERR_UNSUPPORTED;
        }
    }
    if (ctx->tables_valid && memcmp(&ctx->cur_params, p, sizeof(*p)) == 0) return VFSMS_OK;
    static const int dx_s[3][5] = { {0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1} };
    static const int dy_s[3][5] = { {2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1} };
    static const int dxy_s[4][5] = { {1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1} };
    const int lpo = p->n_octave_layers + 2;
    const int nl = lpo * p->n_octaves;
    std::vector<LayerPat> L(nl);
    int step = 1, idx = 0;
    for (int o = 0; o < p->n_octaves; o++) {
        for (int l = 0; l < lpo; l++, idx++) {
            LayerPat &P = L[idx];
            P.size = (9 + 6 * l) << o; P.step = step; P.margin = (P.size / 2) / step; P.octave = o;
            const float ratio = (float)P.size / 9;
            for (int k = 0; k < 10; k++) {
                const int *src = k < 3 ? dx_s[k] : k < 6 ? dy_s[k - 3] : dxy_s[k - 6];
                const int x1 = (int)lrintf(ratio * src[0]), y1 = (int)lrintf(ratio * src[1]);
                const int x2 = (int)lrintf(ratio * src[2]), y2 = (int)lrintf(ratio * src[3]);
                P.box[k][0] = x1; P.box[k][1] = y1; P.box[k][2] = x2; P.box[k][3] = y2;
                for (int c = 0; c < 4; c++)
                    if (P.box[k][c] != vfsms_haar_corner(P.size, k, c)) {       // the LDS Hessian kernels bake these in
                        vfsms_set_error("internal: Haar pattern corner mismatch (size %d box %d)", P.size, k);
                        return VFSMS_ERR_UNSUPPORTED;
                    }
                P.w[k] = src[4] / ((float)(x2 - x1) * (y2 - y1));
            }
        }
        step *= 2;
    }
    SurfTables T;
    memset(&T, 0, sizeof(T));
    float G_ori[13], G_desc[20];
    gaussian_kernel_f32(13, 2.5f, G_ori);
    for (int i = -6; i <= 6; i++)
        for (int j = -6; j <= 6; j++)
            if (i * i + j * j <= 36) {
                T.aptx[T.nOriSamples] = i; T.apty[T.nOriSamples] = j;
                T.aptw[T.nOriSamples++] = G_ori[i + 6] * G_ori[j + 6];
            }
    for (int ang = 0; ang <= 360; ang++)
        for (int w = 0; w < 72; w++) {
            const int d = abs(ang - 5 * w);
            if (d < 30 || d > 330) T.oriMask[ang][w >> 5] |= 1u << (w & 31);
        }
    gaussian_kernel_f32(20, 3.3f, G_desc);
    for (int i = 0; i < 20; i++)
        for (int j = 0; j < 20; j++) T.DW[i * 20 + j] = G_desc[i] * G_desc[j];
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (!ctx->d_layers) HIP_TRY(hipMalloc((void **)&ctx->d_layers, sizeof(LayerPat) * VFSMS_MAX_LAYERS));
    if (!ctx->d_tables) HIP_TRY(hipMalloc((void **)&ctx->d_tables, sizeof(SurfTables)));
    HIP_TRY(hipMemcpy(ctx->d_layers, L.data(), sizeof(LayerPat) * nl, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_tables, &T, sizeof(T), hipMemcpyHostToDevice));
    ctx->n_layers = nl;
    ctx->cur_params = *p;
    ctx->tables_valid = true;
    return VFSMS_OK;
}

// ---- context ---------------------------------------------------------------------------------------------
extern "C" int vfsms_ctx_create(int device, vfsms_ctx **out)
{
    if (!out) return VFSMS_ERR_BAD_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { vfsms_set_error("no HIP device visible"); return VFSMS_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) { vfsms_set_error("device %d out of range (%d visible)", device, n); return VFSMS_ERR_BAD_ARG; }
    HIP_TRY(hipSetDevice(device));
    vfsms_ctx *c = new vfsms_ctx();
    c->device = device; c->arena = nullptr; c->arena_size = 0; c->arena_off = 0;
    c->pinned = nullptr; c->pinned_size = 0; c->pinned_off = 0; c->kp_cap_override = 0;
    c->tables_valid = false; c->d_layers = nullptr; c->d_tables = nullptr; c->n_layers = 0; c->next_handle = 1;
    c->prof_on = false; c->orb_valid = false; c->d_orb_tables = nullptr;
    memset(&c->cur_params, 0, sizeof(c->cur_params));
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { vfsms_set_error("hipStreamCreate: %s", hipGetErrorString(e)); delete c; return VFSMS_ERR_HIP; }
    *out = c;
    return VFSMS_OK;
}

int phase_destroy_plans(vfsms_ctx *ctx);

extern "C" int vfsms_ctx_destroy(vfsms_ctx *ctx)
{
    if (!ctx) return VFSMS_ERR_BAD_ARG;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    phase_destroy_plans(ctx);
    prof_collect(ctx);
    for (hipEvent_t e : ctx->prof_pool) hipEventDestroy(e);
    hipStreamSynchronize(ctx->copy_stream);
    for (auto &kv : ctx->tiles) { if (kv.second.owned) hipFree(kv.second.ptr); if (kv.second.ready) hipEventDestroy(kv.second.ready); }
    for (auto &pe : ctx->tile_pool) { hipFree(pe.ptr); if (pe.idle) hipEventDestroy(pe.idle); }
    for (auto &sb : ctx->stage_pool) hipFree(sb.ptr);
    for (auto &pb : ctx->pin_pool) hipHostFree(pb.ptr);
    for (hipEvent_t ev : ctx->event_pool) hipEventDestroy(ev);
    hipStreamDestroy(ctx->copy_stream);
    if (ctx->stream2) { hipStreamSynchronize(ctx->stream2); hipStreamDestroy(ctx->stream2); }
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    for (auto &kv : ctx->canvases) { hipFree(kv.second.pix); hipFree(kv.second.mask); hipFree(kv.second.d_err); hipFree(kv.second.scratch); }
    if (ctx->has_spare_canvas) { hipFree(ctx->spare_canvas.pix); hipFree(ctx->spare_canvas.mask); hipFree(ctx->spare_canvas.d_err); hipFree(ctx->spare_canvas.scratch); ctx->has_spare_canvas = false; }
    for (auto &kv : ctx->feats) if (!kv.second.block) { if (kv.second.kps_xy) hipFree(kv.second.kps_xy); if (kv.second.desc) hipFree(kv.second.desc); }
    for (auto &kv : ctx->feat_blocks) hipFree(kv.second.base);
    if (ctx->arena) hipFree(ctx->arena);
    if (ctx->pinned) hipHostFree(ctx->pinned);
    if (ctx->d_layers) hipFree(ctx->d_layers);
    if (ctx->d_tables) hipFree(ctx->d_tables);
    if (ctx->d_area_tab) hipFree(ctx->d_area_tab);
    if (ctx->d_orb_tables) hipFree(ctx->d_orb_tables);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return VFSMS_OK;
}
extern "C" int vfsms_ctx_sync(vfsms_ctx *ctx)
{
    if (!ctx) return VFSMS_ERR_BAD_ARG;
    HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VFSMS_OK;
}
extern "C" int vfsms_ctx_sync_uploads(vfsms_ctx *ctx)
{
    if (!ctx) return VFSMS_ERR_BAD_ARG;
    HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
    return VFSMS_OK;
}
extern "C" void *vfsms_ctx_stream(vfsms_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" int vfsms_ctx_set_keypoint_capacity(vfsms_ctx *ctx, int cap)
{
    if (!ctx || cap < 0) return VFSMS_ERR_BAD_ARG;
    ctx->kp_cap_override = cap;
    return VFSMS_OK;
}
static int kp_capacity(vfsms_ctx *ctx, int h, int w)
{
    if (ctx->kp_cap_override > 0) return ctx->kp_cap_override;
    return (int)((long long)h * w / 24) + 4096;
}
#define CTX_ENTER(ctx)                                                     \
    do {                                                                   \
        if (!(ctx)) { vfsms_set_error("null context"); return VFSMS_ERR_BAD_ARG; } \
        HIP_TRY(hipSetDevice((ctx)->device));                              \
    } while (0)

// ---- tiles -------------------------------------------------------------------------------------------------
// A pooled buffer may still be read by work enqueued on the compute stream when its tile was freed (vfsms_canvas_paste_tile and
// vfsms_canvas_fuse_tile_resident without an info readback only enqueue): the stream that writes the buffer next waits for the event
// recorded at vfsms_tile_free.
static int tile_buffer(vfsms_ctx *ctx, size_t bytes, uint8_t **p, hipStream_t writer)
{
    for (size_t k = 0; k < ctx->tile_pool.size(); k++)
        if (ctx->tile_pool[k].bytes == bytes) {
            PoolEnt e = ctx->tile_pool[k];
            ctx->tile_pool.erase(ctx->tile_pool.begin() + k);
            ctx->tile_pool_bytes -= e.bytes;
            if (e.idle) {
                if (writer != ctx->stream) HIP_TRY(hipStreamWaitEvent(writer, e.idle, 0));   // (same-stream reuse is ordered already)
                ctx->event_pool.push_back(e.idle);
            }
            *p = e.ptr;
            return VFSMS_OK;
        }
    HIP_TRY(hipMalloc((void **)p, bytes));
    return VFSMS_OK;
}
// ch > 1: an interleaved colour tile for the mosaic canvas (rows of w * ch bytes; `stride` in bytes); registration takes ch == 1 only
static int tile_upload_impl(vfsms_ctx *ctx, const uint8_t *img, int h, int w_px, int stride, int64_t *handle, bool async, int ch = 1)
{
    const int w = w_px * ch;                                 // bytes per row
    if (!img || !handle || h <= 0 || w_px <= 0 || ch < 1 || ch > 4 || stride < w) { vfsms_set_error("tile_upload: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    TileRec t; t.h = h; t.w = w_px; t.stride = w; t.owned = true; t.ready = nullptr; t.pending = false; t.ch = ch; t.bytes = (size_t)h * w;
    TRY(tile_buffer(ctx, t.bytes, &t.ptr, async ? ctx->copy_stream : ctx->stream));
    if (async) {
        // the copy runs on the context's copy stream; the compute stream waits for it when a batch first names the tile
        if (!ctx->event_pool.empty()) { t.ready = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
        else HIP_TRY(hipEventCreateWithFlags(&t.ready, hipEventDisableTiming));
        HIP_TRY(hipMemcpy2DAsync(t.ptr, w, img, stride, w, h, hipMemcpyHostToDevice, ctx->copy_stream));
        HIP_TRY(hipEventRecord(t.ready, ctx->copy_stream));
        t.pending = true;
    } else {
        HIP_TRY(hipMemcpy2DAsync(t.ptr, w, img, stride, w, h, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    std::lock_guard<std::mutex> lk(ctx->tiles_mu);           // (decoder threads look tiles up concurrently: vfsms_tile_fill)
    *handle = ctx->next_handle++;
    ctx->tiles[*handle] = t;
    return VFSMS_OK;
}
// make the compute stream wait for a tile's asynchronous upload (once)
static int tile_ready(vfsms_ctx *ctx, TileRec &t)
{
    // `fill` and `pending` are written by decoder threads (vfsms_tile_fill*): read them under the same mutex.  Everything else in a
    // TileRec, the tile map's structure, the buffer / event pools and the arena belong to the context's own thread (reserve, upload,
    // free and every batch call must come from it).
    std::unique_lock<std::mutex> lk(ctx->tiles_mu);
    if (t.fill) {                                            // reserved: block until its decoder thread has handed the pixels over
        ctx->tiles_cv.wait(lk, [&] { return t.fill != 1; });
        if (t.fill == 2) { vfsms_set_error("a reserved tile was never filled (its decoder reported a failure)"); return VFSMS_ERR_BAD_ARG; }
    }
    if (t.pending) { HIP_TRY(hipStreamWaitEvent(ctx->stream, t.ready, 0)); t.pending = false; }
    return VFSMS_OK;
}
// non-blocking: has the tile's image been handed over (or was it never a reserved tile)?  An unknown handle counts as ready: the
// evaluator reports it.  (csrc/grid.hip sizes speculative batches by this while decoder threads are still filling tiles.)
int tile_is_filled(vfsms_ctx *ctx, int64_t handle)
{
    std::lock_guard<std::mutex> lk(ctx->tiles_mu);
    auto it = ctx->tiles.find(handle);
    if (it == ctx->tiles.end()) return 1;
    if (it->second.fill == 1) return 0;
    // an asynchronous upload that has not landed yet (vfsms_tile_upload_async: ninety tiles queued on the copy stream in front of a path): the
    // speculative part of a batch takes what is there, like with tiles a decoder still owes -- the first batch does not wait for a whole
    // window of copies
    if (it->second.pending && it->second.ready) {
        const hipError_t e = hipEventQuery(it->second.ready);
        if (e == hipErrorNotReady) return 0;
        if (e != hipSuccess) { (void)hipGetLastError(); return 0; }      // (not left behind as the "last error" of an unrelated launch check)
    }
    return 1;
}
extern "C" int vfsms_tile_upload(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride, int64_t *handle)
{
    CTX_ENTER(ctx);
    return tile_upload_impl(ctx, img, h, w, stride, handle, false);
}
extern "C" int vfsms_tile_upload_async(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride, int64_t *handle)
{
    CTX_ENTER(ctx);
    return tile_upload_impl(ctx, img, h, w, stride, handle, true);
}
extern "C" int vfsms_tile_upload_ch(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int ch, int stride_bytes, int async, int64_t *handle)
{
    CTX_ENTER(ctx);
    return tile_upload_impl(ctx, img, h, w, stride_bytes, handle, async != 0, ch);
}
// ---- tiles whose pixels arrive later, from other threads: the ingest pipeline (Stitcher.py:68-69 decodes file after file BEFORE the
// first pair is registered; here the registration of tiles 0, 1, ... starts while tile k is still being decoded) -------------------------
// Pinned host staging for the decoder threads' hand-overs (vfsms_tile_fill*): the decoder's own memory is pageable, and an asynchronous copy
// from pageable memory makes the runtime pin it on the fly (a trip through the kernel's memory-map lock per tile, contended by every
// decoder thread and by the registrar's launches).  The rows are packed into a pinned buffer by the calling thread (a memcpy, in parallel
// across the decoders) and leave by true DMA.  The pool is shared by the fill threads under stage_mu.
static int stage_pinned_get(vfsms_ctx *ctx, size_t need, StageBuf *out)
{
    {
        std::lock_guard<std::mutex> lk(ctx->stage_mu);
        for (size_t k = 0; k < ctx->pin_pool.size(); k++)
            if (ctx->pin_pool[k].bytes >= need) { *out = ctx->pin_pool[k]; ctx->pin_pool.erase(ctx->pin_pool.begin() + k); return VFSMS_OK; }
    }
    out->ptr = nullptr; out->bytes = need;
    const hipError_t e = hipHostMalloc((void **)&out->ptr, need, hipHostMallocDefault);
    if (e != hipSuccess) {
        // the caller falls back to the pageable hand-over: the runtime's sticky "last error" must not outlive this call, or the launch
        // check of that very fallback (hipGetLastError behind k_ingest_split) would report THIS failure and give the tile up
        (void)hipGetLastError();
        out->ptr = nullptr;
        vfsms_set_error("%s:%d hipHostMalloc(%zu) -> %s", __FILE__, __LINE__, need, hipGetErrorString(e));
        return VFSMS_ERR_HIP;
    }
    return VFSMS_OK;
}
static void stage_pinned_put(vfsms_ctx *ctx, StageBuf b)
{
    if (!b.ptr) return;
    std::lock_guard<std::mutex> lk(ctx->stage_mu);
    if (ctx->pin_pool.size() < 64) ctx->pin_pool.push_back(b); else hipHostFree(b.ptr);
}
static void pack_rows(uint8_t *dst, const uint8_t *src, int stride, size_t row_bytes, int h)
{
    if ((size_t)stride == row_bytes) { memcpy(dst, src, row_bytes * h); return; }
    for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * row_bytes, src + (size_t)y * stride, row_bytes);
}
static int tile_reserve_impl(vfsms_ctx *ctx, int h, int w, int ch, int64_t *handle)
{
    if (!handle || h <= 0 || w <= 0 || ch < 1 || ch > 4) { vfsms_set_error("tile_reserve: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    std::lock_guard<std::mutex> lk(ctx->tiles_mu);
    TileRec t; t.h = h; t.w = w; t.stride = w * ch; t.owned = true; t.ready = nullptr; t.pending = false; t.ch = ch; t.bytes = (size_t)h * w * ch; t.fill = 1;
    TRY(tile_buffer(ctx, t.bytes, &t.ptr, ctx->copy_stream));
    if (!ctx->event_pool.empty()) { t.ready = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
    else {
        hipError_t e = hipEventCreateWithFlags(&t.ready, hipEventDisableTiming);
        if (e != hipSuccess) {                               // the buffer goes back to the pool instead of leaking
            PoolEnt pe; pe.bytes = t.bytes; pe.ptr = t.ptr; pe.idle = nullptr;
            ctx->tile_pool.push_back(pe); ctx->tile_pool_bytes += pe.bytes;
            vfsms_set_error("tile_reserve: %s", hipGetErrorString(e)); return VFSMS_ERR_HIP;
        }
    }
    *handle = ctx->next_handle++;
    ctx->tiles[*handle] = t;
    return VFSMS_OK;
}
extern "C" int vfsms_tile_reserve(vfsms_ctx *ctx, int h, int w, int64_t *handle)
{
    CTX_ENTER(ctx);
    return tile_reserve_impl(ctx, h, w, 1, handle);
}
extern "C" int vfsms_tile_reserve_ch(vfsms_ctx *ctx, int h, int w, int ch, int64_t *handle)
{
    CTX_ENTER(ctx);
    return tile_reserve_impl(ctx, h, w, ch, handle);
}
// May be called from ANY thread, concurrently with batch calls on the context's own thread: copies the pixels on the copy stream and returns
// when the copy has completed, so `img` (a decoder thread's staging buffer) can be reused at once.  img == NULL reports a failed decode:
// the batch call waiting for the tile returns an error instead of waiting forever.  `stride` in bytes; a row is w * ch bytes.
extern "C" int vfsms_tile_fill(vfsms_ctx *ctx, int64_t handle, const uint8_t *img, int stride)
{
    CTX_ENTER(ctx);
    hipEvent_t ev; uint8_t *dst; int h, w;
    {
        std::lock_guard<std::mutex> lk(ctx->tiles_mu);
        auto it = ctx->tiles.find(handle);
        if (it == ctx->tiles.end() || it->second.fill != 1) { vfsms_set_error("tile_fill: not a reserved tile"); return VFSMS_ERR_BAD_ARG; }
        if (!img) { it->second.fill = 2; ctx->tiles_cv.notify_all(); return VFSMS_OK; }
        if (stride < it->second.w * it->second.ch) { vfsms_set_error("tile_fill: stride smaller than a row of the tile"); return VFSMS_ERR_BAD_ARG; }
        ev = it->second.ready; dst = it->second.ptr; h = it->second.h; w = it->second.w * it->second.ch;
    }
    StageBuf pb{nullptr, 0};
    hipError_t e = hipSuccess;
    if (stage_pinned_get(ctx, (size_t)w * h, &pb) == VFSMS_OK) {
        pack_rows(pb.ptr, img, stride, (size_t)w, h);
        e = hipMemcpyAsync(dst, pb.ptr, (size_t)w * h, hipMemcpyHostToDevice, ctx->copy_stream);
    } else e = hipMemcpy2DAsync(dst, w, img, stride, w, h, hipMemcpyHostToDevice, ctx->copy_stream);   // no pinned memory left: straight from the caller's
    if (e == hipSuccess) e = hipEventRecord(ev, ctx->copy_stream);
    if (e == hipSuccess) e = hipEventSynchronize(ev);
    else hipStreamSynchronize(ctx->copy_stream);
    stage_pinned_put(ctx, pb);
    {
        std::lock_guard<std::mutex> lk(ctx->tiles_mu);
        auto it = ctx->tiles.find(handle);
        if (it != ctx->tiles.end()) { it->second.fill = e == hipSuccess ? 0 : 2; it->second.pending = false; }   // (the copy has landed: no stream wait needed)
        ctx->tiles_cv.notify_all();
    }
    if (e != hipSuccess) { vfsms_set_error("tile_fill: %s", hipGetErrorString(e)); return VFSMS_ERR_HIP; }
    return VFSMS_OK;
}
// One decode, both planes (csrc/ingest_kernels.hip): `src` is what the decoder produced ONCE -- format 0: 8-bit gray, 1: Y Cb Cr interleaved,
// 2: Y Cb Cr X (4 bytes per pixel) -- and fills the reserved gray tile `gray` (the registration plane, cv2.imdecode(..., 0) of Stitcher.py:68-69)
// and / or the reserved 3-channel tile `color` (B G R, cv2.imdecode(..., IMREAD_COLOR) of Stitcher.py:382-403); either handle may be 0.
// The source rows go to a device staging buffer on the copy stream, the split / colour conversion runs there too, and the call returns when
// both tiles are complete.  Any thread.  src == NULL gives both tiles up.
int ingest_source_pixel_bytes(int format);
int launch_ingest_split(hipStream_t stream, const uint8_t *src, uint8_t *gray, uint8_t *bgr, long long n, int format);
// the reserved tiles of a pair fill, looked up under the lock: 0 and the geometry, or an error with both tiles left as they were
static int fill_pair_lookup(vfsms_ctx *ctx, int64_t gray, int64_t color, bool give_up, const char *who, int *h, int *w, hipEvent_t *ev, uint8_t **dg, uint8_t **dc)
{
    std::lock_guard<std::mutex> lk(ctx->tiles_mu);
    TileRec *tg = nullptr, *tc = nullptr;
    if (gray) { auto it = ctx->tiles.find(gray); if (it != ctx->tiles.end() && it->second.fill == 1 && it->second.ch == 1) tg = &it->second; }
    if (color) { auto it = ctx->tiles.find(color); if (it != ctx->tiles.end() && it->second.fill == 1 && it->second.ch == 3) tc = &it->second; }
    if ((gray && !tg) || (color && !tc) || (!tg && !tc)) {
        vfsms_set_error("%s: needs a reserved 1-channel tile and / or a reserved 3-channel tile", who); return VFSMS_ERR_BAD_ARG;
    }
    if (give_up) { if (tg) tg->fill = 2; if (tc) tc->fill = 2; ctx->tiles_cv.notify_all(); *h = *w = 0; return VFSMS_OK; }
    if (tg && tc && (tg->h != tc->h || tg->w != tc->w)) { vfsms_set_error("%s: the two tiles differ in size", who); return VFSMS_ERR_BAD_ARG; }
    *h = tg ? tg->h : tc->h; *w = tg ? tg->w : tc->w;
    *ev = tg ? tg->ready : tc->ready; *dg = tg ? tg->ptr : nullptr; *dc = tc ? tc->ptr : nullptr;
    return VFSMS_OK;
}
// `host` (h * w pixels of `format`, densely packed; pinned when `host_pinned`) -> the device staging buffer -> the split kernel -> both tiles
// complete (or both given up, on an error) when this returns
#define VFSMS_SRC_YCC420_RAW 100      // internal: the raw planes of a 4:2:0 JPEG on the iMCU grid (jpeg_decode_raw420_host) -> k_ingest_420
int launch_ingest_420(hipStream_t stream, const uint8_t *src, int pw, int ph, int h, int w, uint8_t *gray, uint8_t *bgr);
static int fill_pair_upload(vfsms_ctx *ctx, int64_t gray, int64_t color, const uint8_t *host, int h, int w, int format, hipEvent_t ev, uint8_t *dg, uint8_t *dc,
                            const char *who)
{
    const int spx = ingest_source_pixel_bytes(format);
    const int pw = (w + 15) & ~15, ph = (h + 15) & ~15;
    const size_t need = format == VFSMS_SRC_YCC420_RAW ? (size_t)pw * ph * 3 / 2 : (size_t)h * w * spx;
    hipError_t e = hipSuccess; int rc = VFSMS_OK;
    StageBuf sb{nullptr, 0};
    if (format == VFSMS_SRC_GRAY8 && !dc) e = hipMemcpyAsync(dg, host, need, hipMemcpyHostToDevice, ctx->copy_stream);      // nothing to convert
    else {
        {   // staging buffer for the packed source rows: a small pool of its own (the tile pool belongs to the context thread)
            std::lock_guard<std::mutex> lk(ctx->stage_mu);
            for (size_t k = 0; k < ctx->stage_pool.size(); k++)
                if (ctx->stage_pool[k].bytes >= need) { sb = ctx->stage_pool[k]; ctx->stage_pool.erase(ctx->stage_pool.begin() + k); break; }
        }
        if (!sb.ptr) { e = hipMalloc((void **)&sb.ptr, need); sb.bytes = need; if (e != hipSuccess) sb.ptr = nullptr; }
        if (e == hipSuccess) e = hipMemcpyAsync(sb.ptr, host, need, hipMemcpyHostToDevice, ctx->copy_stream);
        if (e == hipSuccess) rc = format == VFSMS_SRC_YCC420_RAW ? launch_ingest_420(ctx->copy_stream, sb.ptr, pw, ph, h, w, dg, dc)
                                                                 : launch_ingest_split(ctx->copy_stream, sb.ptr, dg, dc, (long long)h * w, format);
    }
    if (e == hipSuccess && rc == VFSMS_OK) e = hipEventRecord(ev, ctx->copy_stream);
    if (e == hipSuccess && rc == VFSMS_OK) e = hipEventSynchronize(ev);
    if (e != hipSuccess || rc != VFSMS_OK) hipStreamSynchronize(ctx->copy_stream);   // nothing may still read the staging buffers when they are reused
    if (sb.ptr) {
        std::lock_guard<std::mutex> lk(ctx->stage_mu);
        if (ctx->stage_pool.size() < 64) ctx->stage_pool.push_back(sb); else hipFree(sb.ptr);
    }
    const bool ok = e == hipSuccess && rc == VFSMS_OK;
    {
        std::lock_guard<std::mutex> lk(ctx->tiles_mu);
        for (int64_t hd : { gray, color }) {
            if (!hd) continue;
            auto it = ctx->tiles.find(hd);
            if (it != ctx->tiles.end()) { it->second.fill = ok ? 0 : 2; it->second.pending = false; }
        }
        ctx->tiles_cv.notify_all();
    }
    if (e != hipSuccess) { vfsms_set_error("%s: %s", who, hipGetErrorString(e)); return VFSMS_ERR_HIP; }
    return rc;
}
extern "C" int vfsms_tile_fill_pair(vfsms_ctx *ctx, int64_t gray, int64_t color, const uint8_t *src, int stride_bytes, int format)
{
    CTX_ENTER(ctx);
    hipEvent_t ev = nullptr; uint8_t *dg = nullptr, *dc = nullptr; int h = 0, w = 0;
    const int spx = ingest_source_pixel_bytes(format);
    TRY(fill_pair_lookup(ctx, gray, color, src == nullptr, "tile_fill_pair", &h, &w, &ev, &dg, &dc));
    if (!src) return VFSMS_OK;
    if (!spx || stride_bytes < w * spx) { vfsms_set_error("tile_fill_pair: unknown format or stride smaller than a source row"); return VFSMS_ERR_BAD_ARG; }
    const size_t need = (size_t)h * w * spx;
    StageBuf pb{nullptr, 0};
    const uint8_t *host = src;
    std::vector<uint8_t> packed;
    if (stage_pinned_get(ctx, need, &pb) == VFSMS_OK) { pack_rows(pb.ptr, src, stride_bytes, (size_t)w * spx, h); host = pb.ptr; }
    else if ((size_t)stride_bytes != (size_t)w * spx) {      // no pinned memory left: straight from the caller's rows, packed first when they are strided
        pb.ptr = nullptr;
        packed.resize(need); pack_rows(packed.data(), src, stride_bytes, (size_t)w * spx, h); host = packed.data();
    } else pb.ptr = nullptr;
    const int rc = fill_pair_upload(ctx, gray, color, host, h, w, format, ev, dg, dc, "tile_fill_pair");
    stage_pinned_put(ctx, pb);
    return rc;
}

// A JPEG file's bytes -> the reserved gray tile and / or the reserved B G R tile, ONE decode (csrc/jpeg_host.cpp: the system's libjpeg-turbo,
// straight into a pinned staging buffer that is reused from call to call; Y only when no colour tile is asked for, the Y Cb Cr planes
// otherwise, colour conversion on the device).  Any thread; blocks until the tiles are complete.  When the decode cannot be done here
// (VFSMS_ERR_UNSUPPORTED: no libjpeg.so.8 on the host, not a 1- / 3-component JPEG; VFSMS_ERR_BAD_ARG: a damaged file, or a file whose size
// is not the tiles') BOTH TILES STAY RESERVED: the caller decodes some other way and fills them, or gives them up.
int jpeg_decode_host(const unsigned char *jpeg, size_t nbytes, int want_planes, unsigned char *out, size_t cap, int *h_out, int *w_out, int *comp_out);
int jpeg_decode_raw420_host(const unsigned char *jpeg, size_t nbytes, unsigned char *out, size_t cap, int *h_out, int *w_out);
extern "C" int vfsms_tile_fill_jpeg(vfsms_ctx *ctx, int64_t gray, int64_t color, const uint8_t *jpeg, size_t nbytes)
{
    CTX_ENTER(ctx);
    if (!jpeg || !nbytes) { vfsms_set_error("tile_fill_jpeg: no data"); return VFSMS_ERR_BAD_ARG; }
    hipEvent_t ev = nullptr; uint8_t *dg = nullptr, *dc = nullptr; int h = 0, w = 0;
    TRY(fill_pair_lookup(ctx, gray, color, false, "tile_fill_jpeg", &h, &w, &ev, &dg, &dc));
    // (the staging buffer is sized for either form: the interleaved planes are 3 bytes per pixel, the raw 4:2:0 planes 1.5 on the iMCU grid)
    const size_t cap = std::max((size_t)h * w * (dc ? 3 : 1), dc ? (size_t)((w + 15) & ~15) * ((h + 15) & ~15) * 3 / 2 : (size_t)0);
    StageBuf pb{nullptr, 0};
    std::vector<uint8_t> pageable;
    uint8_t *host = nullptr;
    if (stage_pinned_get(ctx, cap, &pb) == VFSMS_OK) host = pb.ptr;
    else { pb.ptr = nullptr; pageable.resize(cap); host = pageable.data(); }
    int jh = 0, jw = 0, comp = 0;
    // Colour wanted and a 4:2:0 file: the host stops behind the IDCT, the device upsamples the chroma and converts (k_ingest_420).
    static const bool raw420 = !(getenv("VFSMS_JPEG_RAW420") && atoi(getenv("VFSMS_JPEG_RAW420")) == 0);
    int rc = VFSMS_ERR_UNSUPPORTED;
    if (dc && raw420) { rc = jpeg_decode_raw420_host(jpeg, nbytes, host, cap, &jh, &jw); comp = 420; }
    if (rc == VFSMS_ERR_UNSUPPORTED) rc = jpeg_decode_host(jpeg, nbytes, dc != nullptr, host, cap, &jh, &jw, &comp);
    if (rc == VFSMS_ERR_CAPACITY || (rc == VFSMS_OK && (jh != h || jw != w))) {
        vfsms_set_error("tile_fill_jpeg: the file is %d x %d, the reserved tiles %d x %d", jh, jw, h, w); rc = VFSMS_ERR_BAD_ARG;
    }
    if (rc == VFSMS_OK)
        rc = fill_pair_upload(ctx, gray, color, host, h, w, comp == 420 ? VFSMS_SRC_YCC420_RAW : comp == 3 ? VFSMS_SRC_YCC24 : VFSMS_SRC_GRAY8, ev, dg, dc, "tile_fill_jpeg");
    stage_pinned_put(ctx, pb);
    return rc;
}

extern "C" int vfsms_host_alloc(vfsms_ctx *ctx, size_t bytes, void **ptr)
{
    CTX_ENTER(ctx);
    if (!ptr || !bytes) { vfsms_set_error("host_alloc: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    HIP_TRY(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return VFSMS_OK;
}
extern "C" int vfsms_host_free(vfsms_ctx *ctx, void *ptr)
{
    CTX_ENTER(ctx);
    if (!ptr) return VFSMS_OK;
    HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
    HIP_TRY(hipHostFree(ptr));
    return VFSMS_OK;
}
extern "C" int vfsms_tile_wrap(vfsms_ctx *ctx, const void *device_ptr, int h, int w, int stride, int64_t *handle)
{
    CTX_ENTER(ctx);
    if (!device_ptr || !handle || h <= 0 || w <= 0 || stride < w) { vfsms_set_error("tile_wrap: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    TileRec t; t.ptr = (uint8_t *)device_ptr; t.h = h; t.w = w; t.stride = stride; t.owned = false; t.ready = nullptr; t.pending = false;
    std::lock_guard<std::mutex> lk(ctx->tiles_mu);
    *handle = ctx->next_handle++;
    ctx->tiles[*handle] = t;
    return VFSMS_OK;
}
extern "C" int vfsms_tile_free(vfsms_ctx *ctx, int64_t handle)
{
    CTX_ENTER(ctx);
    auto it = ctx->tiles.find(handle);
    if (it == ctx->tiles.end()) { vfsms_set_error("tile_free: unknown handle"); return VFSMS_ERR_BAD_ARG; }
    {
        std::lock_guard<std::mutex> lk(ctx->tiles_mu);       // (fill is written by decoder threads)
        if (it->second.fill == 1) { vfsms_set_error("tile_free: the tile is reserved and its decoder has not filled it yet"); return VFSMS_ERR_BAD_ARG; }
    }
    // an upload may still be in flight; compute work on the tile may only be ENQUEUED (canvas paste / resident fuse return early)
    if (it->second.pending) HIP_TRY(hipEventSynchronize(it->second.ready));
    if (it->second.ready) ctx->event_pool.push_back(it->second.ready);
    if (it->second.owned) {
        // the pool is capped by bytes (colour tiles of a mosaic are three times a gray one) and by entries
        if (ctx->tile_pool.size() < 256 && ctx->tile_pool_bytes + it->second.bytes <= ((size_t)8 << 30)) {
            PoolEnt e; e.bytes = it->second.bytes; e.ptr = it->second.ptr; e.idle = nullptr;
            if (!ctx->event_pool.empty()) { e.idle = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
            else HIP_TRY(hipEventCreateWithFlags(&e.idle, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(e.idle, ctx->stream));
            ctx->tile_pool.push_back(e);
            ctx->tile_pool_bytes += e.bytes;
        } else { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(it->second.ptr)); }
    }
    { std::lock_guard<std::mutex> lk(ctx->tiles_mu); ctx->tiles.erase(it); }
    return VFSMS_OK;
}

// ---- helpers ---------------------------------------------------------------------------------------------------
static int upload_image(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride, uint8_t **d)
{
    *d = (uint8_t *)ctx_arena_alloc(ctx, (size_t)h * w);
    if (!*d) { vfsms_set_error("arena exhausted (image upload)"); return VFSMS_ERR_CAPACITY; }
    HIP_TRY(hipMemcpy2DAsync(*d, w, img, stride, w, h, hipMemcpyHostToDevice, ctx->stream));
    return VFSMS_OK;
}
template <typename T>
static int upload_array(vfsms_ctx *ctx, const T *src, size_t n, T **d)
{
    *d = (T *)ctx_arena_alloc(ctx, sizeof(T) * (n ? n : 1));
    if (!*d) { vfsms_set_error("arena exhausted (array upload)"); return VFSMS_ERR_CAPACITY; }
    if (n) HIP_TRY(hipMemcpyAsync(*d, src, sizeof(T) * n, hipMemcpyHostToDevice, ctx->stream));
    return VFSMS_OK;
}

// small host->device uploads of launch records through one pinned staging buffer (a pageable hipMemcpyAsync is staged by
// the runtime and costs a synchronisation each); safe to reuse because every entry point is synchronous at return
int ctx_upload_small(vfsms_ctx *ctx, const void *src, size_t bytes, void **d);
static int upload_pinned(vfsms_ctx *ctx, const void *src, size_t bytes, void **d) { return ctx_upload_small(ctx, src, bytes, d); }
int ctx_upload_small(vfsms_ctx *ctx, const void *src, size_t bytes, void **d)
{
    *d = ctx_arena_alloc(ctx, bytes ? bytes : 1);
    if (!*d) { vfsms_set_error("arena exhausted (record upload)"); return VFSMS_ERR_CAPACITY; }
    if (ctx->pinned_off + bytes > ctx->pinned_size) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));                 // earlier copies out of the old buffer have landed
        if (ctx->pinned) HIP_TRY(hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_size = std::max<size_t>(4 * (ctx->pinned_off + bytes), (size_t)1 << 20);
        HIP_TRY(hipHostMalloc((void **)&ctx->pinned, ctx->pinned_size, hipHostMallocDefault));
        ctx->pinned_off = 0;
    }
    memcpy(ctx->pinned + ctx->pinned_off, src, bytes);
    HIP_TRY(hipMemcpyAsync(*d, ctx->pinned + ctx->pinned_off, bytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->pinned_off += (bytes + 255) & ~(size_t)255;
    return VFSMS_OK;
}

// ---- integral ----------------------------------------------------------------------------------------------------
extern "C" int vfsms_integral_u8_i32(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride, int32_t *sum_out)
{
    CTX_ENTER(ctx);
    if (!img || !sum_out || h <= 0 || w <= 0 || stride < w) { vfsms_set_error("integral: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    const size_t sbytes = sizeof(int32_t) * (size_t)(h + 1) * (w + 1);
    TRY(ctx_arena_reserve(ctx, (size_t)h * w + sbytes + integral_carry_bytes(h, w) + 8192));
    RoiDev R; memset(&R, 0, sizeof(R));
    uint8_t *d_img;
    TRY(upload_image(ctx, img, h, w, stride, &d_img));
    R.img = d_img; R.stride = w; R.h = h; R.w = w;
    R.sum = (int32_t *)ctx_arena_alloc(ctx, sbytes);
    R.ipitch = (w + 3) & ~3;
    R.icarry = (int32_t *)ctx_arena_alloc(ctx, integral_carry_bytes(h, w));
    if (!R.sum || !R.icarry) { vfsms_set_error("arena exhausted (integral)"); return VFSMS_ERR_CAPACITY; }
    RoiDev *d_R;
    TRY(upload_array(ctx, &R, 1, &d_R));
    TRY(launch_integral(ctx, d_R, 1, h, w));
    HIP_TRY(hipMemcpyAsync(sum_out, R.sum, sbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VFSMS_OK;
}

// ---- SURF (host buffers) ------------------------------------------------------------------------------------------
static int surf_host(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride, const vfsms_surf_params *params,
                     float *kps_xy, float *desc, vfsms_keypoint *kps_full, int cap, int *n_out, bool describe)
{
    CTX_ENTER(ctx);
    if (!img || !params || !n_out || h <= 0 || w <= 0 || stride < w || cap < 0) { vfsms_set_error("surf: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    TRY(ctx_prepare_surf(ctx, params));
    const int dim = params->extended ? 128 : 64;
    const int dcap = kp_capacity(ctx, h, w);
    TRY(ctx_arena_reserve(ctx, (size_t)h * w + surf_roi_bytes(h, w, dcap, ctx->n_layers, params->n_octaves, dim) + 65536));
    uint8_t *d_img;
    TRY(upload_image(ctx, img, h, w, stride, &d_img));
    RoiDev R;
    TRY(surf_roi_carve(ctx, &R, d_img, w, h, w, dcap, params));
    RoiDev *d_R;
    TRY(upload_array(ctx, &R, 1, &d_R));
    HIP_TRY(hipMemsetAsync(R.counters, 0, 16 * sizeof(int), ctx->stream));
    TRY(launch_surf_detect(ctx, d_R, &R, 1, params));
    int counters[16];
    if (describe) {
        TRY(launch_surf_describe(ctx, d_R, &R, 1, params));
    }
    HIP_TRY(hipMemcpyAsync(counters, R.counters, sizeof(counters), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (counters[2] || counters[0] > dcap) {
        vfsms_set_error("surf: more than %d keypoint candidates (raise with vfsms_ctx_set_keypoint_capacity)", dcap);
        return VFSMS_ERR_CAPACITY;
    }
    const int n = describe ? counters[1] : counters[0];
    *n_out = n;
    if (n > cap) { vfsms_set_error("surf: %d keypoints exceed the caller's capacity %d", n, cap); return VFSMS_ERR_CAPACITY; }
    if (n > 0) {
        if (describe) {
            if (kps_xy) HIP_TRY(hipMemcpyAsync(kps_xy, R.kps_xy, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, ctx->stream));
            if (desc) HIP_TRY(hipMemcpyAsync(desc, R.desc, sizeof(float) * (size_t)n * dim, hipMemcpyDeviceToHost, ctx->stream));
            if (kps_full) HIP_TRY(hipMemcpyAsync(kps_full, R.kps_out, sizeof(vfsms_keypoint) * n, hipMemcpyDeviceToHost, ctx->stream));
        } else if (kps_full) {
            HIP_TRY(hipMemcpyAsync(kps_full, R.kps, sizeof(vfsms_keypoint) * n, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return VFSMS_OK;
}
extern "C" int vfsms_surf_detect_describe(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride,
                                          const vfsms_surf_params *params, float *kps_xy, float *desc,
                                          vfsms_keypoint *kps_full, int cap, int *n_out)
{
    return surf_host(ctx, img, h, w, stride, params, kps_xy, desc, kps_full, cap, n_out, true);
}
extern "C" int vfsms_surf_detect(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride,
                                 const vfsms_surf_params *params, vfsms_keypoint *kps_full, int cap, int *n_out)
{
    return surf_host(ctx, img, h, w, stride, params, nullptr, nullptr, kps_full, cap, n_out, false);
}

// ---- matching (host buffers) -----------------------------------------------------------------------------------------
static int pick_nsplit(int nq, int nt, int njobs, int dim)
{
    const int qpw = dim == 64 ? 128 : 64;                 // queries per wave
    const long long waves = (long long)((nq + qpw - 1) / qpw) * njobs;
    // aim at >= 16 waves per SIMD (1024 SIMDs) so the last, partially filled round of workgroups is a small tail
    int ns = (int)((16384 + waves - 1) / (waves > 0 ? waves : 1));
    ns = std::max(1, std::min(ns, 64));
    ns = std::min(ns, std::max(1, nt / 64));
    return ns;
}

// train splits of the MFMA filter: a wave holds 64 queries and two waves share a SIMD, so aim at >= 4 rounds of the
// 2048 wave slots; more splits mean more candidate lists to verify, hence the cap
// Round 6: and at least one split per BFM_SPLIT_TRAINS trains.  The candidate lists hold BFM_CAPL = 32 entries per (list, query); what a
// list admits grows with the trains it covers (the threshold window of the bounds pass is an absolute 1.7e-2 in squared distance), and a
// query whose list overflows is verified by an exhaustive scan of ALL trains.  configs[4]'s strips (37 k keypoints, large batches: the
// wave count alone asked for ONE split) spent 11 ms per launch in k_bf_verify_d64 against 0.24 ms at the headline's 8.7 k; with the
// train rule 4.1 ms (one split per 10240 trains; 5120: another 3 % off the search, profiles/r06_ab_bf_split_4096.txt).
#ifndef BFM_SPLIT_TRAINS
#define BFM_SPLIT_TRAINS 5120
#endif
static int pick_filter_nsplit(int nq, int njobs, int nt = 0)
{
    const long long waves = (long long)((nq + 63) / 64) * njobs;
    int ns = (int)((8192 + waves - 1) / (waves > 0 ? waves : 1));
    ns = std::max(ns, (nt + BFM_SPLIT_TRAINS - 1) / BFM_SPLIT_TRAINS);
    return std::max(1, std::min(ns, 8));
}

static bool bf_force_exact()
{
    static const bool v = getenv("VFSMS_BF_EXACT") && atoi(getenv("VFSMS_BF_EXACT")) != 0;
    return v;
}

static int bf_l2_host(vfsms_ctx *ctx, const float *q, int nq, const float *t, int nt, int dim, MatchDev *M, bool with_ratio, double ratio)
{
    const int capq = std::max(nq, 1);
    const int ns_exact = pick_nsplit(nq, nt, 1, dim);
    const int cns = pick_filter_nsplit(nq, 1, nt);
    const bool try_filter = dim == 64 && nq > 0 && nt > 0 && !bf_force_exact();
    TRY(ctx_arena_reserve(ctx, sizeof(float) * ((size_t)nq + nt) * dim + match_bytes(capq, ns_exact) +
                               (try_filter ? match_filter_bytes(capq, std::max(nt, 1), cns) : 0) + 65536));
    memset(M, 0, sizeof(*M));
    float *dq, *dt;
    TRY(upload_array(ctx, q, (size_t)nq * dim, &dq));
    TRY(upload_array(ctx, t, (size_t)nt * dim, &dt));
    int cnt[2] = {nq, nt}; int *dcnt;
    ctx->pinned_off = 0;                                    // entry points are synchronous: the staging buffer is free again
    TRY(upload_pinned(ctx, cnt, sizeof(cnt), (void **)&dcnt));   // copied into pinned staging now: `cnt` may leave scope before the stream runs
    // descriptors of norm <= 1 (SURF's are L2-normalised) take the MFMA-filtered search; anything else the exhaustive kernel
    bool filtered = false;
    if (try_filter) {
        unsigned *d_max = (unsigned *)ctx_arena_alloc(ctx, sizeof(unsigned));
        HIP_TRY(hipMemsetAsync(d_max, 0, sizeof(unsigned), ctx->stream));
        TRY(launch_max_norm2_d64(ctx, dq, nq, d_max));
        TRY(launch_max_norm2_d64(ctx, dt, nt, d_max));
        unsigned bits = 0;
        HIP_TRY(hipMemcpyAsync(&bits, d_max, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        float m; memcpy(&m, &bits, sizeof(m));
        filtered = m <= 1.0001f;
    }
    const int ns = filtered ? 1 : ns_exact;
    TRY(match_carve(ctx, M, capq, dim, ns));
    if (filtered) TRY(match_filter_carve(ctx, M, capq, std::max(nt, 1), cns));
    M->q = dq; M->t = dt; M->nq_ptr = dcnt; M->nt_ptr = dcnt + 1; M->kq = nullptr; M->kt = nullptr;
    MatchDev *dM;
    TRY(upload_array(ctx, M, 1, &dM));
    if (filtered) { TRY(launch_bf_l2_filtered(ctx, dM, 1, capq, std::max(nt, 1), cns)); }
    else { TRY(launch_bf_l2(ctx, dM, 1, capq, ns, dim)); }
    if (with_ratio) {
        TRY(launch_ratio_only(ctx, dM, 1, capq, ratio));
    } else {
        TRY(launch_merge_only(ctx, dM, 1, capq));
    }
    return VFSMS_OK;
}

extern "C" int vfsms_bf_l2_knn2_ratio(vfsms_ctx *ctx, const float *q, int nq, const float *t, int nt, int dim,
                                      double ratio, int32_t *pairs, int cap, int *m_out)
{
    CTX_ENTER(ctx);
    if (!m_out || nq < 0 || nt < 0 || (nq && !q) || (nt && !t)) { vfsms_set_error("bf_l2: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    *m_out = 0;
    if (nq == 0 || nt == 0) return VFSMS_OK;
    MatchDev M;
    TRY(bf_l2_host(ctx, q, nq, t, nt, dim, &M, true, ratio));
    int mc[4];
    HIP_TRY(hipMemcpyAsync(mc, M.mcount, sizeof(mc), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *m_out = mc[0];
    if (mc[0] > cap) { vfsms_set_error("bf_l2: %d matches exceed capacity %d", mc[0], cap); return VFSMS_ERR_CAPACITY; }
    if (mc[0] > 0 && pairs) {
        HIP_TRY(hipMemcpyAsync(pairs, M.pairs, sizeof(int32_t) * 2 * mc[0], hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return VFSMS_OK;
}

extern "C" int vfsms_bf_l2_knn2(vfsms_ctx *ctx, const float *q, int nq, const float *t, int nt, int dim,
                                int32_t *idx1, float *d1, float *d2)
{
    CTX_ENTER(ctx);
    if (nq < 0 || nt < 0 || (nq && !q) || (nt && !t)) { vfsms_set_error("bf_l2_knn2: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    if (nq == 0) return VFSMS_OK;
    if (nt == 0) {
        for (int i = 0; i < nq; i++) { if (idx1) idx1[i] = -1; if (d1) d1[i] = INFINITY; if (d2) d2[i] = INFINITY; }
        return VFSMS_OK;
    }
    MatchDev M;
    TRY(bf_l2_host(ctx, q, nq, t, nt, dim, &M, false, 0.0));
    if (idx1) HIP_TRY(hipMemcpyAsync(idx1, M.i1, sizeof(int) * nq, hipMemcpyDeviceToHost, ctx->stream));
    if (d1) HIP_TRY(hipMemcpyAsync(d1, M.d1, sizeof(float) * nq, hipMemcpyDeviceToHost, ctx->stream));
    if (d2) HIP_TRY(hipMemcpyAsync(d2, M.d2, sizeof(float) * nq, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VFSMS_OK;
}

extern "C" int vfsms_bf_hamming_nn(vfsms_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes,
                                   int max_dist, int32_t *pairs, int cap, int *m_out)
{
    CTX_ENTER(ctx);
    if (!m_out || nq < 0 || nt < 0 || (nq && !q) || (nt && !t)) { vfsms_set_error("bf_hamming: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    *m_out = 0;
    if (nq == 0 || nt == 0) return VFSMS_OK;
    TRY(ctx_arena_reserve(ctx, ((size_t)nq + nt) * nbytes + sizeof(int) * 2 * (size_t)nq + 65536));
    uint8_t *dq, *dt;
    TRY(upload_array(ctx, q, (size_t)nq * nbytes, &dq));
    TRY(upload_array(ctx, t, (size_t)nt * nbytes, &dt));
    int *bi = (int *)ctx_arena_alloc(ctx, sizeof(int) * nq), *bd = (int *)ctx_arena_alloc(ctx, sizeof(int) * nq);
    TRY(launch_bf_hamming(ctx, dq, nq, dt, nt, nbytes, bi, bd));
    std::vector<int> hi(nq), hd(nq);
    HIP_TRY(hipMemcpyAsync(hi.data(), bi, sizeof(int) * nq, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hd.data(), bd, sizeof(int) * nq, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    int m = 0;
    for (int i = 0; i < nq; i++) {
        if (hi[i] < 0) continue;
        if (max_dist >= 0 && !(hd[i] < max_dist)) continue;
        if (m < cap && pairs) { pairs[2 * m] = hi[i]; pairs[2 * m + 1] = i; }
        m++;
    }
    *m_out = m;
    if (m > cap) { vfsms_set_error("bf_hamming: %d matches exceed capacity %d", m, cap); return VFSMS_ERR_CAPACITY; }
    return VFSMS_OK;
}

extern "C" int vfsms_mode_offset(vfsms_ctx *ctx, const float *kpsA, int nA, const float *kpsB, int nB,
                                 const int32_t *pairs, int m, int offset_evaluate, int32_t *out4)
{
    CTX_ENTER(ctx);
    if (!out4 || m < 0 || nA < 0 || nB < 0) { vfsms_set_error("mode_offset: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    if (m == 0) return VFSMS_OK;
    for (int k = 0; k < m; k++)
        if (pairs[2 * k] < 0 || pairs[2 * k] >= nB || pairs[2 * k + 1] < 0 || pairs[2 * k + 1] >= nA) {
            vfsms_set_error("mode_offset: match index out of range"); return VFSMS_ERR_BAD_ARG;
        }
    TRY(ctx_arena_reserve(ctx, sizeof(float) * 2 * ((size_t)nA + nB) + match_bytes(m, 1) + 65536));
    MatchDev M; memset(&M, 0, sizeof(M));
    float *dA, *dB;
    TRY(upload_array(ctx, kpsA, (size_t)2 * nA, &dA));
    TRY(upload_array(ctx, kpsB, (size_t)2 * nB, &dB));
    int cnt[2] = {m, m}; int *dcnt;
    TRY(upload_array(ctx, cnt, 2, &dcnt));
    TRY(match_carve(ctx, &M, m, 64, 1));
    M.kq = dA; M.kt = dB; M.nq_ptr = dcnt; M.nt_ptr = dcnt + 1; M.pairs_given = 1;
    HIP_TRY(hipMemcpyAsync(M.pairs, pairs, sizeof(int32_t) * 2 * m, hipMemcpyHostToDevice, ctx->stream));
    MatchDev *dM;
    TRY(upload_array(ctx, &M, 1, &dM));
    TRY(launch_mode_only(ctx, dM, 1, m, offset_evaluate));
    int32_t res[VFSMS_ATTEMPT_INTS];
    HIP_TRY(hipMemcpyAsync(res, M.result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 4; k++) out4[k] = res[k];
    return VFSMS_OK;
}

// ---- phase correlation ---------------------------------------------------------------------------------------------------
extern "C" int vfsms_phase_correlate_u8(vfsms_ctx *ctx, const uint8_t *a, const uint8_t *b, int h, int w,
                                        int stride_a, int stride_b, double *out3)
{
    CTX_ENTER(ctx);
    if (!a || !b || !out3 || h <= 0 || w <= 0 || stride_a < w || stride_b < w) { vfsms_set_error("phase: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    size_t pb = 0;
    TRY(phase_bytes(ctx, h, w, 1, &pb));
    TRY(ctx_arena_reserve(ctx, 2 * (size_t)h * w + pb + 65536));
    ctx->pinned_off = 0;
    uint8_t *da, *db;
    TRY(upload_image(ctx, a, h, w, stride_a, &da));
    TRY(upload_image(ctx, b, h, w, stride_b, &db));
    double *d_out = (double *)ctx_arena_alloc(ctx, 3 * sizeof(double));
    TRY(phase_correlate_device(ctx, da, w, db, w, h, w, d_out));
    HIP_TRY(hipMemcpyAsync(out3, d_out, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VFSMS_OK;
}

static int resolve_job(vfsms_ctx *ctx, const vfsms_roi_pair &j, const uint8_t **pa, int *sa, const uint8_t **pb, int *sb)
{
    auto ia = ctx->tiles.find(j.tile_a), ib = ctx->tiles.find(j.tile_b);
    if (ia == ctx->tiles.end() || ib == ctx->tiles.end()) { vfsms_set_error("attempt: unknown tile handle"); return VFSMS_ERR_BAD_ARG; }
    TileRec &A = ia->second, &B = ib->second;
    if (A.ch != 1 || B.ch != 1) { vfsms_set_error("attempt: registration takes single-channel tiles"); return VFSMS_ERR_BAD_ARG; }
    TRY(tile_ready(ctx, A)); TRY(tile_ready(ctx, B));
    if (j.h <= 0 || j.w <= 0 || j.ay0 < 0 || j.ax0 < 0 || j.by0 < 0 || j.bx0 < 0 || j.ay0 + j.h > A.h || j.ax0 + j.w > A.w ||
        j.by0 + j.h > B.h || j.bx0 + j.w > B.w) { vfsms_set_error("attempt: ROI outside its tile"); return VFSMS_ERR_BAD_ARG; }
    *pa = A.ptr + (size_t)j.ay0 * A.stride + j.ax0; *sa = A.stride;
    *pb = B.ptr + (size_t)j.by0 * B.stride + j.bx0; *sb = B.stride;
    return VFSMS_OK;
}

extern "C" int vfsms_attempt_phase_batch(vfsms_ctx *ctx, const vfsms_roi_pair *jobs, int n, double *out)
{
    CTX_ENTER(ctx);
    if (n < 0 || (n && (!jobs || !out))) { vfsms_set_error("attempt_phase: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    if (n == 0) return VFSMS_OK;
    // attempts of one ROI size (all of them, in practice) run as ONE batched transform; other sizes follow group by group
    std::vector<int> order(n);
    for (int k = 0; k < n; k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
        return jobs[x].h != jobs[y].h ? jobs[x].h < jobs[y].h : jobs[x].w < jobs[y].w; });
    size_t need = 0;
    for (int g0 = 0; g0 < n;) {
        int g1 = g0;
        while (g1 < n && jobs[order[g1]].h == jobs[order[g0]].h && jobs[order[g1]].w == jobs[order[g0]].w) g1++;
        size_t pb = 0;
        TRY(phase_bytes(ctx, jobs[order[g0]].h, jobs[order[g0]].w, g1 - g0, &pb));
        need = std::max(need, pb);
        g0 = g1;
    }
    TRY(ctx_arena_reserve(ctx, need + sizeof(double) * 3 * n + 65536));
    ctx->pinned_off = 0;
    double *d_out = (double *)ctx_arena_alloc(ctx, sizeof(double) * 3 * n);   // in group order; un-permuted on the host
    const size_t mark = ctx->arena_off;
    std::vector<PhaseJobHost> pj(n);
    for (int g0 = 0; g0 < n;) {
        int g1 = g0;
        while (g1 < n && jobs[order[g1]].h == jobs[order[g0]].h && jobs[order[g1]].w == jobs[order[g0]].w) g1++;
        for (int k = g0; k < g1; k++) {
            const uint8_t *pa, *pb; int sa, sb;
            TRY(resolve_job(ctx, jobs[order[k]], &pa, &sa, &pb, &sb));
            pj[k].a = pa; pj[k].b = pb; pj[k].sa = sa; pj[k].sb = sb;
        }
        ctx->arena_off = mark;                               // stream order makes scratch reuse safe
        TRY(phase_correlate_batch_device(ctx, pj.data() + g0, g1 - g0, jobs[order[g0]].h, jobs[order[g0]].w, d_out + 3 * g0));
        g0 = g1;
    }
    std::vector<double> tmp((size_t)3 * n);
    HIP_TRY(hipMemcpyAsync(tmp.data(), d_out, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < n; k++) for (int c = 0; c < 3; c++) out[3 * order[k] + c] = tmp[3 * k + c];
    return VFSMS_OK;
}

// ---- fused SURF + BF + ratio + mode attempts --------------------------------------------------------------------------------
// the second compute stream and its fork / join events (created on first use)
static int ctx_second_stream(vfsms_ctx *ctx)
{
    if (ctx->stream2) return VFSMS_OK;
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    return VFSMS_OK;
}
// ctx->stream is what every launcher enqueues on: swapped for a scope, restored on every way out of it
struct StreamSwap {
    vfsms_ctx *c; hipStream_t saved;
    StreamSwap(vfsms_ctx *ctx, hipStream_t s) : c(ctx), saved(ctx->stream) { ctx->stream = s; }
    ~StreamSwap() { c->stream = saved; }
};
// Joins the second stream on EVERY way out of the forked region of attempt_surf_impl: an error return behind the fork (e.g. a capacity
// overflow of part 1's describe) would otherwise let the caller's retry reset and reuse the arena while part 0's search still runs on it.
struct SecondStreamJoin {
    vfsms_ctx *c; bool armed = false;
    explicit SecondStreamJoin(vfsms_ctx *ctx) : c(ctx) {}
    ~SecondStreamJoin() { if (armed && c->stream2) (void)hipStreamSynchronize(c->stream2); }
};

static int attempt_surf_impl(vfsms_ctx *ctx, const vfsms_roi_pair *jobs, int n, const vfsms_surf_params *params, double ratio,
                             int offset_evaluate, int enh_mode, double clip_limit, int tile_grid, int32_t *out)
{
    CTX_ENTER(ctx);
    if (n < 0 || (n && (!jobs || !out)) || !params) { vfsms_set_error("attempt_surf: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    if (n == 0) return VFSMS_OK;
    TRY(ctx_prepare_surf(ctx, params));
    const int dim = params->extended ? 128 : 64;
    size_t need = 0; int maxcap = 0;
    std::vector<int> caps(n);
    for (int k = 0; k < n; k++) {
        caps[k] = kp_capacity(ctx, jobs[k].h, jobs[k].w);
        maxcap = std::max(maxcap, caps[k]);
    }
    // 64-d descriptors leave the descriptor kernel with norm <= 1: their 2-NN search runs as an MFMA candidate filter plus
    // exact verification (match_kernels.hip); other widths, or VFSMS_BF_EXACT=1, take the exhaustive VALU kernel.
    const bool filtered = dim == 64 && !bf_force_exact();
    const int cns = pick_filter_nsplit(maxcap * 2 / 3, n, maxcap * 2 / 3);   // the registrar sizes the capacity at 1.5x the largest ROI seen
    const int ns = filtered ? 1 : pick_nsplit(maxcap / 3, maxcap / 3, n, dim);   // typical occupancy of the capacity
    for (int k = 0; k < n; k++)
        need += 2 * surf_roi_bytes(jobs[k].h, jobs[k].w, caps[k], ctx->n_layers, params->n_octaves, dim) + match_bytes(caps[k], ns) +
                (filtered ? match_filter_bytes(caps[k], caps[k], cns) : 0);
    need += (sizeof(RoiDev) * 2 + sizeof(MatchDev)) * n + 64 * 3 * n + 65536;
    if (enh_mode) for (int k = 0; k < n; k++) need += 2 * enhance_scratch_bytes(jobs[k].h, jobs[k].w, enh_mode, tile_grid) + 2 * sizeof(EnhJob) + 512;
    TRY(ctx_arena_reserve(ctx, need));
    ctx->pinned_off = 0;
    std::vector<RoiDev> R(2 * n);
    std::vector<EnhJob> E(enh_mode ? 2 * n : 0);
    std::vector<MatchDev> M(n);
    // counters of all ROIs and results of all jobs live in two contiguous blocks: one memset, two D2H copies per batch
    int *cblock = (int *)ctx_arena_alloc(ctx, sizeof(int) * 16 * 2 * n);
    int32_t *rblock = (int32_t *)ctx_arena_alloc(ctx, sizeof(int32_t) * VFSMS_ATTEMPT_INTS * n);
    // ROIs of one shape next to each other (the column strips, then the strips of the turn candidates): the kernels whose grid follows the
    // image size are launched per shape run (surf_kernels.hip: shape_runs).  Slot s of the batch holds job ord[s]; results go to the job's row.
    std::vector<int> ord(n);
    for (int k = 0; k < n; k++) ord[k] = k;
    std::stable_sort(ord.begin(), ord.end(), [&](int a_, int b_) {
        return jobs[a_].h != jobs[b_].h ? jobs[a_].h < jobs[b_].h : jobs[a_].w < jobs[b_].w; });
    for (int s_ = 0; s_ < n; s_++) {
        const int k = ord[s_];
        const uint8_t *pa, *pb; int sa, sb;
        TRY(resolve_job(ctx, jobs[k], &pa, &sa, &pb, &sb));
        if (enh_mode) {            // Stitcher.py:327-334: the ROI strips are equalised / CLAHE'd before detectAndDescribe
            TRY(enhance_carve(ctx, &E[2 * s_], pa, sa, jobs[k].h, jobs[k].w, enh_mode, tile_grid));
            TRY(enhance_carve(ctx, &E[2 * s_ + 1], pb, sb, jobs[k].h, jobs[k].w, enh_mode, tile_grid));
            pa = E[2 * s_].dst; sa = jobs[k].w; pb = E[2 * s_ + 1].dst; sb = jobs[k].w;
        }
        TRY(surf_roi_carve(ctx, &R[2 * s_], pa, sa, jobs[k].h, jobs[k].w, caps[k], params));
        TRY(surf_roi_carve(ctx, &R[2 * s_ + 1], pb, sb, jobs[k].h, jobs[k].w, caps[k], params));
        R[2 * s_].counters = cblock + 16 * (2 * s_); R[2 * s_ + 1].counters = cblock + 16 * (2 * s_ + 1);
        memset(&M[s_], 0, sizeof(MatchDev));
        TRY(match_carve(ctx, &M[s_], caps[k], dim, ns));
        if (filtered) TRY(match_filter_carve(ctx, &M[s_], caps[k], caps[k], cns));
        M[s_].result = rblock + VFSMS_ATTEMPT_INTS * k;
        M[s_].q = R[2 * s_].desc; M[s_].t = R[2 * s_ + 1].desc;
        M[s_].nq_ptr = R[2 * s_].counters + 1; M[s_].nt_ptr = R[2 * s_ + 1].counters + 1;
        M[s_].kq = R[2 * s_].kps_xy; M[s_].kt = R[2 * s_ + 1].kps_xy;
    }
    RoiDev *dR; MatchDev *dM;
    TRY(upload_pinned(ctx, R.data(), sizeof(RoiDev) * 2 * n, (void **)&dR));
    TRY(upload_pinned(ctx, M.data(), sizeof(MatchDev) * n, (void **)&dM));
    if (enh_mode) {
        EnhJob *dE;
        TRY(upload_pinned(ctx, E.data(), sizeof(EnhJob) * 2 * n, (void **)&dE));
        TRY(launch_enhance(ctx, dE, E.data(), 2 * n, enh_mode, clip_limit, tile_grid));
    }
    HIP_TRY(hipMemsetAsync(cblock, 0, sizeof(int) * 16 * 2 * n, ctx->stream));
    // Two pipes at once -- built, measured, OFF by default.  The 2-NN search lives on the matrix cores (k_bf_mfma16_d64: MFMA pipe 55-65 %
    // busy, VALU idle), detection on the VALU and the texture-address path.  With VFSMS_OVERLAP=1 a large batch is cut in two parts of
    // slots: part 0 (VFSMS_OVERLAP_PCT, default 70 %) is detected and described, then its search + ratio + vote run on the second stream
    // BESIDE the detect stage of part 1 (the persistent descriptor kernel takes every CU's LDS, so what can overlap is integral / Hessian /
    // NMS / sort / orientation of part 1).  Same kernels, same results (the GPU suite passes either way).  On the bench (A B A B, one call,
    // profiles/r05_ab_overlap.txt): 52.03 ms per step on one stream, 52.85 with the overlap -- the second stream's 5.5 ms of stages do run
    // concurrently (stage sum 57.4 ms against 52.8 ms of wall clock), but Hessian and integral slow down by what the search takes from
    // them (8.95 vs 6.85 ms, 1.25 vs 0.38 ms) and the halved launches add their tails: these kernels fill the chip on their own, a second
    // queue only re-divides it.  (Rounds 2-3 found the same for two half batches of the same mix.)  Round 6: the search beside DESCRIBE
    // instead cannot happen at all -- k_describe holds 6 x 80 of a SIMD's 512 VGPRs, a k_bf_mfma16_d64<1> wave needs 168.
    static const bool overlap_on = getenv("VFSMS_OVERLAP") && atoi(getenv("VFSMS_OVERLAP")) != 0;
    static const int overlap_pct = getenv("VFSMS_OVERLAP_PCT") ? atoi(getenv("VFSMS_OVERLAP_PCT")) : 70;
    const int n0 = (overlap_on && filtered && n >= 12) ? std::min(n - 2, std::max(2, n * overlap_pct / 100)) : n;
    if (n0 < n) TRY(ctx_second_stream(ctx));
    TRY(launch_surf_detect(ctx, dR, R.data(), 2 * n0, params));
    TRY(launch_surf_describe(ctx, dR, R.data(), 2 * n0, params));
    SecondStreamJoin join_guard(ctx);
    if (n0 < n) {
        HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        join_guard.armed = true;
        {
            StreamSwap on_second(ctx, ctx->stream2);              // the launchers enqueue on ctx->stream (their profiling events too)
            TRY(launch_bf_l2_filtered(ctx, dM, n0, maxcap, maxcap, cns));
            TRY(launch_ratio_mode(ctx, dM, n0, maxcap, ratio, offset_evaluate));
            HIP_TRY(hipEventRecord(ctx->ev_join, ctx->stream2));
        }
        TRY(launch_surf_detect(ctx, dR + 2 * n0, R.data() + 2 * n0, 2 * (n - n0), params));
        TRY(launch_surf_describe(ctx, dR + 2 * n0, R.data() + 2 * n0, 2 * (n - n0), params));
        TRY(launch_bf_l2_filtered(ctx, dM + n0, n - n0, maxcap, maxcap, cns));
        TRY(launch_ratio_mode(ctx, dM + n0, n - n0, maxcap, ratio, offset_evaluate));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        join_guard.armed = false;                        // joined in stream order: the synchronisation below covers both streams
    } else {
        if (filtered) { TRY(launch_bf_l2_filtered(ctx, dM, n, maxcap, maxcap, cns)); }
        else { TRY(launch_bf_l2(ctx, dM, n, maxcap, ns, dim)); }
        TRY(launch_ratio_mode(ctx, dM, n, maxcap, ratio, offset_evaluate));
    }
    std::vector<int> counters((size_t)16 * 2 * n);
    HIP_TRY(hipMemcpyAsync(counters.data(), cblock, sizeof(int) * 16 * 2 * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, rblock, sizeof(int32_t) * VFSMS_ATTEMPT_INTS * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 2 * n; k++)
        if (counters[(size_t)16 * k + 2] || counters[(size_t)16 * k] > R[k].cap) {
            vfsms_set_error("attempt_surf: ROI %d exceeded %d keypoint candidates (vfsms_ctx_set_keypoint_capacity)", k, R[k].cap);
            return VFSMS_ERR_CAPACITY;
        }
    return VFSMS_OK;
}

extern "C" int vfsms_attempt_surf_batch(vfsms_ctx *ctx, const vfsms_roi_pair *jobs, int n,
                                        const vfsms_surf_params *params, double ratio, int offset_evaluate, int32_t *out)
{
    return attempt_surf_impl(ctx, jobs, n, params, ratio, offset_evaluate, 0, 0.0, 0, out);
}
extern "C" int vfsms_attempt_surf_batch_enhanced(vfsms_ctx *ctx, const vfsms_roi_pair *jobs, int n, const vfsms_surf_params *params,
                                                 double ratio, int offset_evaluate, int enhance_mode, double clip_limit, int tile_grid,
                                                 int32_t *out)
{
    if (enhance_mode < 0 || enhance_mode > 2) { vfsms_set_error("attempt_surf: enhance_mode must be 0, 1 or 2"); return VFSMS_ERR_BAD_ARG; }
    return attempt_surf_impl(ctx, jobs, n, params, ratio, offset_evaluate, enhance_mode, clip_limit, tile_grid, out);
}

// ---- enhancement (host buffers) ---------------------------------------------------------------------------------------------------
extern "C" int vfsms_enhance_u8(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride, int mode, double clip_limit, int tile_grid,
                                uint8_t *out)
{
    CTX_ENTER(ctx);
    if (!img || !out || h <= 0 || w <= 0 || stride < w || mode < 1 || mode > 2) { vfsms_set_error("enhance: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    TRY(ctx_arena_reserve(ctx, (size_t)h * w + enhance_scratch_bytes(h, w, mode, tile_grid) + 65536));
    ctx->pinned_off = 0;
    uint8_t *d_img;
    TRY(upload_image(ctx, img, h, w, stride, &d_img));
    EnhJob J, *dJ;
    TRY(enhance_carve(ctx, &J, d_img, w, h, w, mode, tile_grid));
    TRY(upload_pinned(ctx, &J, sizeof(J), (void **)&dJ));
    TRY(launch_enhance(ctx, dJ, &J, 1, mode, clip_limit, tile_grid));
    HIP_TRY(hipMemcpyAsync(out, J.dst, (size_t)h * w, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VFSMS_OK;
}

// ---- device-resident feature sets: the payload of Stitcher.tempImageFeature (Stitcher.py:14-18, 278-290) -----------------------------
extern "C" int vfsms_features_surf(vfsms_ctx *ctx, int64_t tile, int y0, int x0, int h, int w, const vfsms_surf_params *params,
                                   int enhance_mode, double clip_limit, int tile_grid, int64_t *feat, int *n_out)
{
    CTX_ENTER(ctx);
    auto it = ctx->tiles.find(tile);
    if (it == ctx->tiles.end() || !params || !feat || !n_out) { vfsms_set_error("features_surf: bad arguments / unknown tile"); return VFSMS_ERR_BAD_ARG; }
    TileRec &T = it->second;
    if (T.ch != 1) { vfsms_set_error("features_surf: registration takes single-channel tiles"); return VFSMS_ERR_BAD_ARG; }
    TRY(tile_ready(ctx, T));
    if (h <= 0 || w <= 0 || y0 < 0 || x0 < 0 || y0 + h > T.h || x0 + w > T.w) { vfsms_set_error("features_surf: ROI outside the tile"); return VFSMS_ERR_BAD_ARG; }
    if (enhance_mode < 0 || enhance_mode > 2) { vfsms_set_error("features_surf: enhance_mode must be 0, 1 or 2"); return VFSMS_ERR_BAD_ARG; }
    TRY(ctx_prepare_surf(ctx, params));
    const int dim = params->extended ? 128 : 64;
    const int dcap = kp_capacity(ctx, h, w);
    TRY(ctx_arena_reserve(ctx, surf_roi_bytes(h, w, dcap, ctx->n_layers, params->n_octaves, dim) +
                               (enhance_mode ? enhance_scratch_bytes(h, w, enhance_mode, tile_grid) : 0) + 65536));
    ctx->pinned_off = 0;
    const uint8_t *src = T.ptr + (size_t)y0 * T.stride + x0;
    int sstride = T.stride;
    if (enhance_mode) {
        EnhJob J, *dJ;
        TRY(enhance_carve(ctx, &J, src, sstride, h, w, enhance_mode, tile_grid));
        TRY(upload_pinned(ctx, &J, sizeof(J), (void **)&dJ));
        TRY(launch_enhance(ctx, dJ, &J, 1, enhance_mode, clip_limit, tile_grid));
        src = J.dst; sstride = w;
    }
    RoiDev R, *d_R;
    TRY(surf_roi_carve(ctx, &R, src, sstride, h, w, dcap, params));
    TRY(upload_pinned(ctx, &R, sizeof(R), (void **)&d_R));
    HIP_TRY(hipMemsetAsync(R.counters, 0, 16 * sizeof(int), ctx->stream));
    TRY(launch_surf_detect(ctx, d_R, &R, 1, params));
    TRY(launch_surf_describe(ctx, d_R, &R, 1, params));
    int counters[16];
    HIP_TRY(hipMemcpyAsync(counters, R.counters, sizeof(counters), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (counters[2] || counters[0] > dcap) {
        vfsms_set_error("features_surf: more than %d keypoint candidates (raise with vfsms_ctx_set_keypoint_capacity)", dcap);
        return VFSMS_ERR_CAPACITY;
    }
    FeatRec F; F.n = counters[1]; F.dim = dim; F.is_orb = 0; F.kps_xy = nullptr; F.desc = nullptr;
    if (F.n > 0) {
        HIP_TRY(hipMalloc((void **)&F.kps_xy, sizeof(float) * 2 * F.n));
        HIP_TRY(hipMalloc(&F.desc, sizeof(float) * (size_t)F.n * dim));
        HIP_TRY(hipMemcpyAsync(F.kps_xy, R.kps_xy, sizeof(float) * 2 * F.n, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(F.desc, R.desc, sizeof(float) * (size_t)F.n * dim, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    *feat = ctx->next_handle++;
    ctx->feats[*feat] = F;
    *n_out = F.n;
    return VFSMS_OK;
}

// Whole-tile feature sets of MANY tiles in fused launches: the line scans of Main.py:29-51 (4 of the 6 demo datasets) call
// calculateOffsetForFeatureSearch pair after pair (Stitcher.py:260-304); with all tiles of a scan in HBM every tile is described once, up to
// 16 per launch sequence, and the N - 1 matches + mode votes run as one batch (vfsms_features_match_offset_batch) -- one host synchronisation
// per 16 tiles instead of two per tile.
static int features_surf_batch_impl(vfsms_ctx *ctx, const int64_t *tiles, int n, const vfsms_surf_params *params,
                                    int enhance_mode, double clip_limit, int tile_grid, int64_t *feats, int *counts);
extern "C" int vfsms_features_free(vfsms_ctx *ctx, int64_t feat);
extern "C" int vfsms_features_surf_batch(vfsms_ctx *ctx, const int64_t *tiles, int n, const vfsms_surf_params *params,
                                         int enhance_mode, double clip_limit, int tile_grid, int64_t *feats, int *counts)
{
    // the output array is cleared BEFORE anything can fail: the clean-up below frees every live handle it finds in it, and a caller's array
    // may still hold handles of an earlier call (they are the caller's; an early error -- bad arguments, unsupported parameters -- must not
    // release them)
    if (feats && n > 0) for (int k = 0; k < n; k++) feats[k] = 0;
    const int rc = features_surf_batch_impl(ctx, tiles, n, params, enhance_mode, clip_limit, tile_grid, feats, counts);
    if (rc != VFSMS_OK && ctx && feats && n > 0) {
        // an error in chunk 2 or later (e.g. VFSMS_ERR_CAPACITY): the sets of the earlier chunks never reach the caller -- release them
        // (and their shared block) here; the error text of the failure is kept
        char msg[512]; vfsms_last_error(msg, sizeof(msg));
        for (int k = 0; k < n; k++)
            if (feats[k] && ctx->feats.count(feats[k])) { vfsms_features_free(ctx, feats[k]); feats[k] = 0; }
        vfsms_set_error("%s", msg);
    }
    return rc;
}
static int features_surf_batch_impl(vfsms_ctx *ctx, const int64_t *tiles, int n, const vfsms_surf_params *params,
                                    int enhance_mode, double clip_limit, int tile_grid, int64_t *feats, int *counts)
{
    CTX_ENTER(ctx);
    if (n < 0 || (n && (!tiles || !feats || !counts)) || !params) { vfsms_set_error("features_surf_batch: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    if (enhance_mode < 0 || enhance_mode > 2) { vfsms_set_error("features_surf_batch: enhance_mode must be 0, 1 or 2"); return VFSMS_ERR_BAD_ARG; }
    TRY(ctx_prepare_surf(ctx, params));
    const int dim = params->extended ? 128 : 64;
    for (int k = 0; k < n; k++) feats[k] = 0;
    for (int c0 = 0; c0 < n;) {
        // a chunk: at most 16 tiles and ~6 GB of scratch
        int c1 = c0; size_t need = 0;
        std::vector<TileRec *> T; std::vector<int> caps;
        while (c1 < n && c1 - c0 < 16) {
            auto it = ctx->tiles.find(tiles[c1]);
            if (it == ctx->tiles.end()) { vfsms_set_error("features_surf_batch: unknown tile handle"); return VFSMS_ERR_BAD_ARG; }
            TileRec &t = it->second;
            if (t.ch != 1) { vfsms_set_error("features_surf_batch: registration takes single-channel tiles"); return VFSMS_ERR_BAD_ARG; }
            const int cap = kp_capacity(ctx, t.h, t.w);
            const size_t b = surf_roi_bytes(t.h, t.w, cap, ctx->n_layers, params->n_octaves, dim) +
                             (enhance_mode ? enhance_scratch_bytes(t.h, t.w, enhance_mode, tile_grid) + sizeof(EnhJob) + 512 : 0);
            if (c1 > c0 && need + b > ((size_t)6 << 30)) break;
            need += b; T.push_back(&t); caps.push_back(cap); c1++;
        }
        const int m = c1 - c0;
        TRY(ctx_arena_reserve(ctx, need + (sizeof(RoiDev) + 64) * m + 65536));
        ctx->pinned_off = 0;
        std::vector<RoiDev> R(m);
        std::vector<EnhJob> E(enhance_mode ? m : 0);
        int *cblock = (int *)ctx_arena_alloc(ctx, sizeof(int) * 16 * m);
        for (int k = 0; k < m; k++) {
            TRY(tile_ready(ctx, *T[k]));
            const uint8_t *src = T[k]->ptr; int sstride = T[k]->stride;
            if (enhance_mode) {
                TRY(enhance_carve(ctx, &E[k], src, sstride, T[k]->h, T[k]->w, enhance_mode, tile_grid));
                src = E[k].dst; sstride = T[k]->w;
            }
            TRY(surf_roi_carve(ctx, &R[k], src, sstride, T[k]->h, T[k]->w, caps[k], params));
            R[k].counters = cblock + 16 * k;
        }
        RoiDev *dR;
        TRY(upload_pinned(ctx, R.data(), sizeof(RoiDev) * m, (void **)&dR));
        if (enhance_mode) {
            EnhJob *dE;
            TRY(upload_pinned(ctx, E.data(), sizeof(EnhJob) * m, (void **)&dE));
            TRY(launch_enhance(ctx, dE, E.data(), m, enhance_mode, clip_limit, tile_grid));
        }
        HIP_TRY(hipMemsetAsync(cblock, 0, sizeof(int) * 16 * m, ctx->stream));
        TRY(launch_surf_detect(ctx, dR, R.data(), m, params));
        TRY(launch_surf_describe(ctx, dR, R.data(), m, params));
        std::vector<int> counters((size_t)16 * m);
        HIP_TRY(hipMemcpyAsync(counters.data(), cblock, sizeof(int) * 16 * m, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < m; k++)
            if (counters[(size_t)16 * k + 2] || counters[(size_t)16 * k] > R[k].cap) {
                vfsms_set_error("features_surf_batch: tile %d exceeded %d keypoint candidates (vfsms_ctx_set_keypoint_capacity)", c0 + k, R[k].cap);
                return VFSMS_ERR_CAPACITY;
            }
        // ONE allocation for the sets of the chunk (a hipMalloc / hipFree pair per set costs more than describing it)
        size_t total = 0;
        for (int k = 0; k < m; k++) total += (((size_t)counters[(size_t)16 * k + 1] * (2 + dim) * sizeof(float)) + 255) & ~(size_t)255;
        char *base = nullptr; int64_t blk = 0;
        if (total) {
            HIP_TRY(hipMalloc((void **)&base, total));
            blk = ctx->next_handle++;
            ctx->feat_blocks[blk] = FeatBlock{base, 0};
        }
        size_t off = 0;
        for (int k = 0; k < m; k++) {
            FeatRec F; F.n = counters[(size_t)16 * k + 1]; F.dim = dim; F.is_orb = 0; F.kps_xy = nullptr; F.desc = nullptr;
            if (F.n > 0) {
                F.block = blk; ctx->feat_blocks[blk].refs++;
                F.kps_xy = (float *)(base + off); F.desc = base + off + sizeof(float) * 2 * F.n;
                off += (((size_t)F.n * (2 + dim) * sizeof(float)) + 255) & ~(size_t)255;
                HIP_TRY(hipMemcpyAsync(F.kps_xy, R[k].kps_xy, sizeof(float) * 2 * F.n, hipMemcpyDeviceToDevice, ctx->stream));
                HIP_TRY(hipMemcpyAsync(F.desc, R[k].desc, sizeof(float) * (size_t)F.n * dim, hipMemcpyDeviceToDevice, ctx->stream));
            }
            feats[c0 + k] = ctx->next_handle++;
            ctx->feats[feats[c0 + k]] = F;
            counts[c0 + k] = F.n;
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));          // the copies out of the arena have landed before the next chunk reuses it
        c0 = c1;
    }
    return VFSMS_OK;
}

// matchDescriptors + getOffsetByMode of n (query set A_k, train set B_k) jobs as ONE batch: out[8 * k ..] as vfsms_features_match_offset
extern "C" int vfsms_features_match_offset_batch(vfsms_ctx *ctx, const int64_t *feat_a, const int64_t *feat_b, int n, double ratio,
                                                 int offset_evaluate, int32_t *out)
{
    CTX_ENTER(ctx);
    if (n < 0 || (n && (!feat_a || !feat_b || !out))) { vfsms_set_error("features_match_batch: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    std::vector<const FeatRec *> A(n), B(n);
    std::vector<int> live;
    int maxq = 0, maxt = 0, dim = 0;
    for (int k = 0; k < n; k++) {
        auto ia = ctx->feats.find(feat_a[k]), ib = ctx->feats.find(feat_b[k]);
        if (ia == ctx->feats.end() || ib == ctx->feats.end()) { vfsms_set_error("features_match_batch: unknown handle"); return VFSMS_ERR_BAD_ARG; }
        A[k] = &ia->second; B[k] = &ib->second;
        if (A[k]->dim != B[k]->dim || A[k]->is_orb || B[k]->is_orb || (dim && A[k]->dim != dim)) { vfsms_set_error("features_match_batch: descriptor kinds differ"); return VFSMS_ERR_BAD_ARG; }
        dim = A[k]->dim;
        for (int c = 0; c < VFSMS_ATTEMPT_INTS; c++) out[VFSMS_ATTEMPT_INTS * k + c] = 0;
        out[VFSMS_ATTEMPT_INTS * k + 4] = A[k]->n; out[VFSMS_ATTEMPT_INTS * k + 5] = B[k]->n;
        if (A[k]->n > 0 && B[k]->n > 0) { live.push_back(k); maxq = std::max(maxq, A[k]->n); maxt = std::max(maxt, B[k]->n); }
    }
    const int m = (int)live.size();
    if (m == 0) return VFSMS_OK;
    const bool filtered = dim == 64 && !bf_force_exact();
    const int cns = pick_filter_nsplit(maxq, m, maxt);
    const int ns = filtered ? 1 : pick_nsplit(maxq, maxt, m, dim);
    size_t need = 0;
    for (int j = 0; j < m; j++)
        need += match_bytes(A[live[j]]->n, ns) + (filtered ? match_filter_bytes(A[live[j]]->n, B[live[j]]->n, cns) : 0);
    TRY(ctx_arena_reserve(ctx, need + (sizeof(MatchDev) + 64 + sizeof(int32_t) * VFSMS_ATTEMPT_INTS) * m + 65536));
    ctx->pinned_off = 0;
    std::vector<MatchDev> M(m);
    std::vector<int> cnt(2 * m);
    for (int j = 0; j < m; j++) { cnt[2 * j] = A[live[j]]->n; cnt[2 * j + 1] = B[live[j]]->n; }
    int32_t *rblock = (int32_t *)ctx_arena_alloc(ctx, sizeof(int32_t) * VFSMS_ATTEMPT_INTS * m);
    int *dcnt;
    TRY(upload_pinned(ctx, cnt.data(), sizeof(int) * 2 * m, (void **)&dcnt));
    for (int j = 0; j < m; j++) {
        const FeatRec &a = *A[live[j]], &b = *B[live[j]];
        memset(&M[j], 0, sizeof(MatchDev));
        TRY(match_carve(ctx, &M[j], a.n, dim, ns));
        if (filtered) TRY(match_filter_carve(ctx, &M[j], a.n, b.n, cns));
        M[j].result = rblock + VFSMS_ATTEMPT_INTS * j;
        M[j].q = (const float *)a.desc; M[j].t = (const float *)b.desc; M[j].kq = a.kps_xy; M[j].kt = b.kps_xy;
        M[j].nq_ptr = dcnt + 2 * j; M[j].nt_ptr = dcnt + 2 * j + 1;
    }
    MatchDev *dM;
    TRY(upload_pinned(ctx, M.data(), sizeof(MatchDev) * m, (void **)&dM));
    if (filtered) { TRY(launch_bf_l2_filtered(ctx, dM, m, maxq, maxt, cns)); }
    else { TRY(launch_bf_l2(ctx, dM, m, maxq, ns, dim)); }
    TRY(launch_ratio_mode(ctx, dM, m, maxq, ratio, offset_evaluate));
    std::vector<int32_t> res((size_t)VFSMS_ATTEMPT_INTS * m);
    HIP_TRY(hipMemcpyAsync(res.data(), rblock, sizeof(int32_t) * VFSMS_ATTEMPT_INTS * m, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < m; j++)
        for (int c = 0; c < VFSMS_ATTEMPT_INTS; c++) out[VFSMS_ATTEMPT_INTS * live[j] + c] = res[(size_t)VFSMS_ATTEMPT_INTS * j + c];
    return VFSMS_OK;
}

extern "C" int vfsms_features_free(vfsms_ctx *ctx, int64_t feat)
{
    CTX_ENTER(ctx);
    auto it = ctx->feats.find(feat);
    if (it == ctx->feats.end()) { vfsms_set_error("features_free: unknown handle"); return VFSMS_ERR_BAD_ARG; }
    if (it->second.block) {
        auto bt = ctx->feat_blocks.find(it->second.block);
        if (bt != ctx->feat_blocks.end() && --bt->second.refs == 0) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipFree(bt->second.base));
            ctx->feat_blocks.erase(bt);
        }
    } else {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (it->second.kps_xy) HIP_TRY(hipFree(it->second.kps_xy));
        if (it->second.desc) HIP_TRY(hipFree(it->second.desc));
    }
    ctx->feats.erase(it);
    return VFSMS_OK;
}

extern "C" int vfsms_features_download(vfsms_ctx *ctx, int64_t feat, float *kps_xy, float *desc, int cap, int *n_out, int *dim_out)
{
    CTX_ENTER(ctx);
    auto it = ctx->feats.find(feat);
    if (it == ctx->feats.end() || !n_out) { vfsms_set_error("features_download: unknown handle"); return VFSMS_ERR_BAD_ARG; }
    const FeatRec &F = it->second;
    *n_out = F.n;
    if (dim_out) *dim_out = F.dim;
    if (F.n > cap) { vfsms_set_error("features_download: %d keypoints exceed the caller's capacity %d", F.n, cap); return VFSMS_ERR_CAPACITY; }
    if (F.n > 0) {
        if (kps_xy) HIP_TRY(hipMemcpyAsync(kps_xy, F.kps_xy, sizeof(float) * 2 * F.n, hipMemcpyDeviceToHost, ctx->stream));
        if (desc) HIP_TRY(hipMemcpyAsync(desc, F.desc, sizeof(float) * (size_t)F.n * F.dim, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return VFSMS_OK;
}

// matchDescriptors + getOffsetByMode on two resident sets (query = A, train = B): out[8] as in vfsms_attempt_surf_batch
extern "C" int vfsms_features_match_offset(vfsms_ctx *ctx, int64_t feat_a, int64_t feat_b, double ratio, int offset_evaluate, int32_t *out)
{
    CTX_ENTER(ctx);
    auto ia = ctx->feats.find(feat_a), ib = ctx->feats.find(feat_b);
    if (ia == ctx->feats.end() || ib == ctx->feats.end() || !out) { vfsms_set_error("features_match: unknown handle"); return VFSMS_ERR_BAD_ARG; }
    const FeatRec &A = ia->second, &B = ib->second;
    if (A.dim != B.dim || A.is_orb != B.is_orb) { vfsms_set_error("features_match: descriptor kinds differ"); return VFSMS_ERR_BAD_ARG; }
    for (int k = 0; k < VFSMS_ATTEMPT_INTS; k++) out[k] = 0;
    out[4] = A.n; out[5] = B.n;
    if (A.n == 0 || B.n == 0) return VFSMS_OK;
    const int dim = A.dim, capq = A.n;
    const bool filtered = dim == 64 && !bf_force_exact();          // SURF descriptors are L2-normalised by construction
    const int cns = pick_filter_nsplit(A.n, 1, B.n);
    const int ns = filtered ? 1 : pick_nsplit(A.n, B.n, 1, dim);
    TRY(ctx_arena_reserve(ctx, match_bytes(capq, ns) + (filtered ? match_filter_bytes(capq, B.n, cns) : 0) + 65536));
    ctx->pinned_off = 0;
    MatchDev M; memset(&M, 0, sizeof(M));
    TRY(match_carve(ctx, &M, capq, dim, ns));
    if (filtered) TRY(match_filter_carve(ctx, &M, capq, B.n, cns));
    int cnt[2] = {A.n, B.n}; int *dcnt;
    TRY(upload_pinned(ctx, cnt, sizeof(cnt), (void **)&dcnt));
    M.q = (const float *)A.desc; M.t = (const float *)B.desc; M.kq = A.kps_xy; M.kt = B.kps_xy; M.nq_ptr = dcnt; M.nt_ptr = dcnt + 1;
    MatchDev *dM;
    TRY(upload_pinned(ctx, &M, sizeof(M), (void **)&dM));
    if (filtered) { TRY(launch_bf_l2_filtered(ctx, dM, 1, capq, B.n, cns)); }
    else { TRY(launch_bf_l2(ctx, dM, 1, capq, ns, dim)); }
    TRY(launch_ratio_mode(ctx, dM, 1, capq, ratio, offset_evaluate));
    HIP_TRY(hipMemcpyAsync(out, M.result, sizeof(int32_t) * VFSMS_ATTEMPT_INTS, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VFSMS_OK;
}

// ---- fuse ----------------------------------------------------------------------------------------------------------------------
extern "C" int vfsms_fuse_fade_i64(vfsms_ctx *ctx, const int64_t *A, const int64_t *B, int r, int c, int ch,
                                   int dx, int dy, uint8_t *out, int32_t *info)
{
    CTX_ENTER(ctx);
    if (!A || !B || !out || r <= 0 || c <= 0 || ch < 1 || ch > 4) { vfsms_set_error("fuse_i64: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    const size_t nel = (size_t)r * c * ch;
    TRY(ctx_arena_reserve(ctx, nel * 17 + sizeof(float) * 4 * ((size_t)r + c) + sizeof(int) * 4 * ((size_t)r + c) + 65536));
    long long *dA, *dB;
    TRY(upload_array(ctx, (const long long *)A, nel, &dA));
    TRY(upload_array(ctx, (const long long *)B, nel, &dB));
    uint8_t *d_out = (uint8_t *)ctx_arena_alloc(ctx, nel);
    TRY(fuse_i64_device(ctx, dA, dB, r, c, ch, dx, dy, d_out, info));
    HIP_TRY(hipMemcpyAsync(out, d_out, nel, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VFSMS_OK;
}

// ImageFusion.fuseByTrigonometric (ImageFusion.py:246-293) on the reference's own array representation
extern "C" int vfsms_fuse_trig_i64(vfsms_ctx *ctx, const int64_t *A, const int64_t *B, int r, int c, int ch,
                                   int dx, int dy, uint8_t *out, int32_t *info)
{
    CTX_ENTER(ctx);
    if (!A || !B || !out || r <= 0 || c <= 0 || ch < 1 || ch > 4) { vfsms_set_error("fuse_trig_i64: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    const size_t nel = (size_t)r * c * ch;
    TRY(ctx_arena_reserve(ctx, nel * 17 + sizeof(float) * 4 * ((size_t)r + c) + sizeof(int) * 4 * ((size_t)r + c) + 65536));
    long long *dA, *dB;
    TRY(upload_array(ctx, (const long long *)A, nel, &dA));
    TRY(upload_array(ctx, (const long long *)B, nel, &dB));
    uint8_t *d_out = (uint8_t *)ctx_arena_alloc(ctx, nel);
    TRY(fuse_i64_device(ctx, dA, dB, r, c, ch, dx, dy, d_out, info, 1));
    HIP_TRY(hipMemcpyAsync(out, d_out, nel, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VFSMS_OK;
}

int fuse_i64_ramps(vfsms_ctx *ctx, const long long *dA, int r, int c, int ch, int dx, int dy, int force_corner,
                   float *h_ramps, int32_t *info);

extern "C" int vfsms_fuse_ramps_i64(vfsms_ctx *ctx, const int64_t *A, int r, int c, int ch, int dx, int dy,
                                    int force_corner, float *ramps, int32_t *info)
{
    CTX_ENTER(ctx);
    if (!A || !ramps || r <= 0 || c <= 0 || ch < 1 || ch > 4) { vfsms_set_error("fuse_ramps: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    const size_t nel = (size_t)r * c * ch;
    TRY(ctx_arena_reserve(ctx, nel * 8 + sizeof(float) * 8 * ((size_t)r + c) + 65536));
    long long *dA;
    TRY(upload_array(ctx, (const long long *)A, nel, &dA));
    TRY(fuse_i64_ramps(ctx, dA, r, c, ch, dx, dy, force_corner, ramps, info));
    return VFSMS_OK;
}

extern "C" int vfsms_canvas_create(vfsms_ctx *ctx, int rows, int cols, int ch, int64_t *handle)
{
    CTX_ENTER(ctx);
    if (!handle || rows <= 0 || cols <= 0 || ch < 1 || ch > 4) { vfsms_set_error("canvas_create: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    CanvasRec cv; cv.rows = rows; cv.cols = cols; cv.ch = ch;
    if (ctx->has_spare_canvas && ctx->spare_canvas.rows == rows && ctx->spare_canvas.cols == cols && ctx->spare_canvas.ch == ch) {
        cv = ctx->spare_canvas; cv.placed.clear();          // same size as the canvas freed last: its buffers, re-initialised below in stream order
        ctx->has_spare_canvas = false;
    } else {
        HIP_TRY(hipMalloc((void **)&cv.pix, (size_t)rows * cols * ch));
        HIP_TRY(hipMalloc((void **)&cv.mask, (size_t)rows * cols));
        HIP_TRY(hipMalloc((void **)&cv.d_err, sizeof(int)));
        HIP_TRY(hipMalloc(&cv.scratch, canvas_scratch_bytes(rows, cols)));
    }
    TRY(canvas_scratch_init(ctx, &cv));
    HIP_TRY(hipMemsetAsync(cv.d_err, 0, sizeof(int), ctx->stream));
    HIP_TRY(hipMemsetAsync(cv.pix, 0, (size_t)rows * cols * ch, ctx->stream));
    HIP_TRY(hipMemsetAsync(cv.mask, 0, (size_t)rows * cols, ctx->stream));
    *handle = ctx->next_handle++;
    ctx->canvases[*handle] = cv;
    return VFSMS_OK;
}
extern "C" int vfsms_canvas_free(vfsms_ctx *ctx, int64_t handle)
{
    CTX_ENTER(ctx);
    auto it = ctx->canvases.find(handle);
    if (it == ctx->canvases.end()) { vfsms_set_error("canvas_free: unknown handle"); return VFSMS_ERR_BAD_ARG; }
    if (ctx->has_spare_canvas) {                               // one spare at a time: the older one goes back to the device
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        CanvasRec &o = ctx->spare_canvas;
        HIP_TRY(hipFree(o.pix)); HIP_TRY(hipFree(o.mask)); HIP_TRY(hipFree(o.d_err)); HIP_TRY(hipFree(o.scratch));
    }
    ctx->spare_canvas = it->second; ctx->has_spare_canvas = true;      // kept for a canvas of the same size (stream order makes the reuse safe)
    ctx->canvases.erase(it);
    return VFSMS_OK;
}
static int canvas_tile_args(vfsms_ctx *ctx, int64_t canvas, const uint8_t *tile, int h, int w, int y0, int x0, CanvasRec **cv)
{
    auto it = ctx->canvases.find(canvas);
    if (it == ctx->canvases.end()) { vfsms_set_error("canvas: unknown handle"); return VFSMS_ERR_BAD_ARG; }
    *cv = &it->second;
    if (!tile || h <= 0 || w <= 0 || y0 < 0 || x0 < 0 || y0 + h > (*cv)->rows || x0 + w > (*cv)->cols) {
        vfsms_set_error("canvas: tile rectangle outside the canvas"); return VFSMS_ERR_BAD_ARG;
    }
    return VFSMS_OK;
}
extern "C" int vfsms_canvas_paste(vfsms_ctx *ctx, int64_t canvas, const uint8_t *tile, int h, int w, int y0, int x0)
{
    CTX_ENTER(ctx);
    CanvasRec *cv;
    TRY(canvas_tile_args(ctx, canvas, tile, h, w, y0, x0, &cv));
    const size_t nb = (size_t)h * w * cv->ch;
    TRY(ctx_arena_reserve(ctx, nb + 65536));
    uint8_t *d_tile;
    TRY(upload_array(ctx, tile, nb, &d_tile));
    TRY(canvas_paste_device(ctx, cv, d_tile, h, w, y0, x0));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VFSMS_OK;
}
extern "C" int vfsms_canvas_fuse_tile_m(vfsms_ctx *ctx, int64_t canvas, const uint8_t *tile, int h, int w,
                                        int y0, int x0, int ry0, int rx0, int ry1, int rx1, int dx, int dy, int method, int32_t *info)
{
    CTX_ENTER(ctx);
    if (method < 0 || method > 1) { vfsms_set_error("canvas_fuse_tile: method must be 0 (fadeInAndFadeOut) or 1 (trigonometric)"); return VFSMS_ERR_BAD_ARG; }
    CanvasRec *cv;
    TRY(canvas_tile_args(ctx, canvas, tile, h, w, y0, x0, &cv));
    if (ry1 > ry0 && rx1 > rx0 && (ry0 < y0 || rx0 < x0 || ry1 > y0 + h || rx1 > x0 + w)) {
        vfsms_set_error("canvas_fuse_tile: fuse ROI must lie inside the tile rectangle"); return VFSMS_ERR_BAD_ARG;
    }
    const size_t nb = (size_t)h * w * cv->ch;
    const int r = std::max(ry1 - ry0, 0), c = std::max(rx1 - rx0, 0);
    TRY(ctx_arena_reserve(ctx, nb + sizeof(float) * 8 * ((size_t)r + c) + 65536));
    uint8_t *d_tile;
    TRY(upload_array(ctx, tile, nb, &d_tile));
    TRY(canvas_fuse_device(ctx, cv, d_tile, h, w, y0, x0, ry0, rx0, ry1, rx1, dx, dy, info, method));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VFSMS_OK;
}
extern "C" int vfsms_canvas_fuse_tile(vfsms_ctx *ctx, int64_t canvas, const uint8_t *tile, int h, int w,
                                      int y0, int x0, int ry0, int rx0, int ry1, int rx1, int dx, int dy, int32_t *info)
{
    return vfsms_canvas_fuse_tile_m(ctx, canvas, tile, h, w, y0, x0, ry0, rx0, ry1, rx1, dx, dy, 0, info);
}
extern "C" int vfsms_canvas_blend_tile(vfsms_ctx *ctx, int64_t canvas, const uint8_t *tile, int h, int w,
                                       int y0, int x0, int ry0, int rx0, int ry1, int rx1, int mode)
{
    CTX_ENTER(ctx);
    CanvasRec *cv;
    TRY(canvas_tile_args(ctx, canvas, tile, h, w, y0, x0, &cv));
    if (mode < 0 || mode > 2) { vfsms_set_error("canvas_blend_tile: mode must be 0 (average), 1 (maximum) or 2 (minimum)"); return VFSMS_ERR_BAD_ARG; }
    if (ry1 > ry0 && rx1 > rx0 && (ry0 < y0 || rx0 < x0 || ry1 > y0 + h || rx1 > x0 + w)) {
        vfsms_set_error("canvas_blend_tile: fuse ROI must lie inside the tile rectangle"); return VFSMS_ERR_BAD_ARG;
    }
    const size_t nb = (size_t)h * w * cv->ch;
    TRY(ctx_arena_reserve(ctx, nb + 65536));
    uint8_t *d_tile;
    TRY(upload_array(ctx, tile, nb, &d_tile));
    TRY(canvas_blend_device(ctx, cv, d_tile, h, w, y0, x0, ry0, rx0, ry1, rx1, mode));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VFSMS_OK;
}
static int canvas_resident_args(vfsms_ctx *ctx, int64_t canvas, int64_t tile, int y0, int x0, CanvasRec **cv, TileRec **tr)
{
    auto it = ctx->canvases.find(canvas);
    auto jt = ctx->tiles.find(tile);
    if (it == ctx->canvases.end() || jt == ctx->tiles.end()) { vfsms_set_error("canvas: unknown canvas or tile handle"); return VFSMS_ERR_BAD_ARG; }
    *cv = &it->second; *tr = &jt->second;
    TRY(tile_ready(ctx, jt->second));
    if ((*cv)->ch != (*tr)->ch || (*tr)->stride != (*tr)->w * (*tr)->ch) {
        vfsms_set_error("canvas: a resident tile must have the canvas's channel count and be densely packed (stride == w * ch)"); return VFSMS_ERR_BAD_ARG;
    }
    if (y0 < 0 || x0 < 0 || y0 + (*tr)->h > (*cv)->rows || x0 + (*tr)->w > (*cv)->cols) {
        vfsms_set_error("canvas: tile rectangle outside the canvas"); return VFSMS_ERR_BAD_ARG;
    }
    return VFSMS_OK;
}
extern "C" int vfsms_canvas_paste_tile(vfsms_ctx *ctx, int64_t canvas, int64_t tile, int y0, int x0)
{
    CTX_ENTER(ctx);
    CanvasRec *cv; TileRec *tr;
    TRY(canvas_resident_args(ctx, canvas, tile, y0, x0, &cv, &tr));
    TRY(canvas_paste_device(ctx, cv, tr->ptr, tr->h, tr->w, y0, x0));
    return VFSMS_OK;                                       // enqueued only: resident tiles need no host synchronisation
}
extern "C" int vfsms_canvas_fuse_tile_resident_m(vfsms_ctx *ctx, int64_t canvas, int64_t tile,
                                                 int y0, int x0, int ry0, int rx0, int ry1, int rx1, int dx, int dy, int method, int32_t *info)
{
    CTX_ENTER(ctx);
    if (method < 0 || method > 1) { vfsms_set_error("canvas_fuse_tile: method must be 0 (fadeInAndFadeOut) or 1 (trigonometric)"); return VFSMS_ERR_BAD_ARG; }
    CanvasRec *cv; TileRec *tr;
    TRY(canvas_resident_args(ctx, canvas, tile, y0, x0, &cv, &tr));
    const int h = tr->h, w = tr->w;
    if (ry1 > ry0 && rx1 > rx0 && (ry0 < y0 || rx0 < x0 || ry1 > y0 + h || rx1 > x0 + w)) {
        vfsms_set_error("canvas_fuse_tile: fuse ROI must lie inside the tile rectangle"); return VFSMS_ERR_BAD_ARG;
    }
    const int r = std::max(ry1 - ry0, 0), c = std::max(rx1 - rx0, 0);
    TRY(ctx_arena_reserve(ctx, sizeof(float) * 8 * ((size_t)r + c) + 65536));
    TRY(canvas_fuse_device(ctx, cv, tr->ptr, h, w, y0, x0, ry0, rx0, ry1, rx1, dx, dy, info, method));
    if (info) HIP_TRY(hipStreamSynchronize(ctx->stream));   // without a readback the call only enqueues (stream order keeps the canvas consistent)
    return VFSMS_OK;
}
// fuseMethod "average" / "maximum" / "minimum" (mode 0 / 1 / 2) with a resident tile; enqueue only
extern "C" int vfsms_canvas_blend_tile_resident(vfsms_ctx *ctx, int64_t canvas, int64_t tile,
                                                int y0, int x0, int ry0, int rx0, int ry1, int rx1, int mode)
{
    CTX_ENTER(ctx);
    if (mode < 0 || mode > 2) { vfsms_set_error("canvas_blend_tile: mode must be 0 (average), 1 (maximum) or 2 (minimum)"); return VFSMS_ERR_BAD_ARG; }
    CanvasRec *cv; TileRec *tr;
    TRY(canvas_resident_args(ctx, canvas, tile, y0, x0, &cv, &tr));
    if (ry1 > ry0 && rx1 > rx0 && (ry0 < y0 || rx0 < x0 || ry1 > y0 + tr->h || rx1 > x0 + tr->w)) {
        vfsms_set_error("canvas_blend_tile: fuse ROI must lie inside the tile rectangle"); return VFSMS_ERR_BAD_ARG;
    }
    TRY(canvas_blend_device(ctx, cv, tr->ptr, tr->h, tr->w, y0, x0, ry0, rx0, ry1, rx1, mode));
    return VFSMS_OK;
}
extern "C" int vfsms_canvas_fuse_tile_resident(vfsms_ctx *ctx, int64_t canvas, int64_t tile,
                                               int y0, int x0, int ry0, int rx0, int ry1, int rx1, int dx, int dy, int32_t *info)
{
    return vfsms_canvas_fuse_tile_resident_m(ctx, canvas, tile, y0, x0, ry0, rx0, ry1, rx1, dx, dy, 0, info);
}
// The whole mosaic walk of Stitcher.getStitchByOffset (Stitcher.py:434-483) over resident tiles as ONE call: per tile nine ints
// [y0, x0, ry0, rx0, ry1, rx1, dx, dy, mode] with mode -1 = paste (the first tile, notFuse), 0 = fadeInAndFadeOut, 1 = trigonometric,
// 2 / 3 / 4 = average / maximum / minimum.
// Enqueue only (one library call per mosaic instead of one per tile; the device chain stays two launches per tile); geometry errors are latched
// in the canvas and reported by the download, as with vfsms_canvas_fuse_tile_resident(info = NULL).
extern "C" int vfsms_canvas_assemble_resident(vfsms_ctx *ctx, int64_t canvas, int n, const int64_t *tiles, const int32_t *geom)
{
    CTX_ENTER(ctx);
    if (n < 0 || (n > 0 && (!tiles || !geom))) { vfsms_set_error("canvas_assemble_resident: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    for (int i = 0; i < n; i++) {                       // everything is checked before anything is enqueued
        const int32_t *g = geom + 9 * (size_t)i;
        if (g[8] < -1 || g[8] > 4) { vfsms_set_error("canvas_assemble_resident: mode must be -1 (paste), 0 (fadeInAndFadeOut), 1 (trigonometric), 2 / 3 / 4 (average / maximum / minimum)"); return VFSMS_ERR_BAD_ARG; }
        CanvasRec *cv; TileRec *tr;
        TRY(canvas_resident_args(ctx, canvas, tiles[i], g[0], g[1], &cv, &tr));
        if (g[8] >= 0 && g[4] > g[2] && g[5] > g[3] && (g[2] < g[0] || g[3] < g[1] || g[4] > g[0] + tr->h || g[5] > g[1] + tr->w)) {
            vfsms_set_error("canvas_assemble_resident: fuse ROI must lie inside the tile rectangle"); return VFSMS_ERR_BAD_ARG;
        }
    }
    for (int i = 0; i < n; i++) {
        const int32_t *g = geom + 9 * (size_t)i;
        if (g[8] < 0) TRY(vfsms_canvas_paste_tile(ctx, canvas, tiles[i], g[0], g[1]));
        else if (g[8] >= 2) TRY(vfsms_canvas_blend_tile_resident(ctx, canvas, tiles[i], g[0], g[1], g[2], g[3], g[4], g[5], g[8] - 2));
        else TRY(vfsms_canvas_fuse_tile_resident_m(ctx, canvas, tiles[i], g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], nullptr));
    }
    return VFSMS_OK;
}
extern "C" int vfsms_canvas_download(vfsms_ctx *ctx, int64_t canvas, uint8_t *out)
{
    CTX_ENTER(ctx);
    auto it = ctx->canvases.find(canvas);
    if (it == ctx->canvases.end() || !out) { vfsms_set_error("canvas_download: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    const CanvasRec &cv = it->second;
    // never-written pixels are still 0 (the canvas is zero-initialised), exactly Stitcher.py:485
    int err = 0;
    HIP_TRY(hipMemcpyAsync(&err, cv.d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, cv.pix, (size_t)cv.rows * cv.cols * cv.ch, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (err) {
        vfsms_set_error("fuse: degenerate corner geometry in one of the fused tiles (the reference's getWeightsMatrix raises there)");
        return VFSMS_ERR_BAD_ARG;
    }
    return VFSMS_OK;
}

// rows [row0, row0 + nrows) of the canvas: a multi-GB mosaic leaves the device band by band (streamed write-out, Stitcher.py:174-179)
extern "C" int vfsms_canvas_download_rows(vfsms_ctx *ctx, int64_t canvas, int row0, int nrows, uint8_t *out)
{
    CTX_ENTER(ctx);
    auto it = ctx->canvases.find(canvas);
    if (it == ctx->canvases.end() || !out || row0 < 0 || nrows <= 0 || row0 + nrows > it->second.rows) {
        vfsms_set_error("canvas_download_rows: bad arguments"); return VFSMS_ERR_BAD_ARG;
    }
    const CanvasRec &cv = it->second;
    const size_t pitch = (size_t)cv.cols * cv.ch;
    int err = 0;
    HIP_TRY(hipMemcpyAsync(&err, cv.d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, cv.pix + (size_t)row0 * pitch, (size_t)nrows * pitch, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (err) {
        vfsms_set_error("fuse: degenerate corner geometry in one of the fused tiles (the reference's getWeightsMatrix raises there)");
        return VFSMS_ERR_BAD_ARG;
    }
    return VFSMS_OK;
}

// ---- ORB -------------------------------------------------------------------------------------------------------------------------
static void orb_caps(const vfsms_orb_params *p, int *cap1, int *cap2, int *cap)
{
    // level-0 quota bounds every level; FAST scores are integers, so ties can exceed 2 x quota -- leave generous room
    const float factor = 1.f / p->scale_factor;
    const int q0 = (int)lrintf(p->n_features * (1 - factor) / (1 - powf(factor, (float)p->n_levels)));
    *cap1 = 4 * q0 + 2048; *cap2 = 2 * q0 + 1024; *cap = 2 * p->n_features + 2048;
}

extern "C" int vfsms_orb_detect_describe(vfsms_ctx *ctx, const uint8_t *img, int h, int w, int stride,
                                         const vfsms_orb_params *params, float *kps_xy, uint8_t *desc,
                                         vfsms_keypoint *kps_full, int cap, int *n_out)
{
    CTX_ENTER(ctx);
    if (!img || !params || !n_out || h <= 0 || w <= 0 || stride < w || cap < 0) { vfsms_set_error("orb: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    TRY(ctx_prepare_orb(ctx, params));
    int c1, c2, c;
    orb_caps(params, &c1, &c2, &c);
    TRY(ctx_arena_reserve(ctx, (size_t)h * w + orb_roi_bytes(params, h, w, c1, c2, c) + 65536));
    uint8_t *d_img;
    TRY(upload_image(ctx, img, h, w, stride, &d_img));
    OrbDev R;
    TRY(orb_roi_carve(ctx, &R, d_img, w, h, w, params, c1, c2, c));
    OrbDev *dR;
    TRY(upload_array(ctx, &R, 1, &dR));
    TRY(launch_orb(ctx, dR, &R, 1, params));
    int counters[16];
    HIP_TRY(hipMemcpyAsync(counters, R.counters, sizeof(counters), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (counters[2]) { vfsms_set_error("orb: internal keypoint capacity exceeded"); return VFSMS_ERR_CAPACITY; }
    const int n = counters[1];
    *n_out = n;
    if (n > cap) { vfsms_set_error("orb: %d keypoints exceed the caller's capacity %d", n, cap); return VFSMS_ERR_CAPACITY; }
    if (n > 0) {
        if (kps_xy) HIP_TRY(hipMemcpyAsync(kps_xy, R.kps_xy, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, ctx->stream));
        if (desc) HIP_TRY(hipMemcpyAsync(desc, R.desc, (size_t)32 * n, hipMemcpyDeviceToHost, ctx->stream));
        if (kps_full) HIP_TRY(hipMemcpyAsync(kps_full, R.kps_out, sizeof(vfsms_keypoint) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return VFSMS_OK;
}

extern "C" int vfsms_attempt_orb_batch(vfsms_ctx *ctx, const vfsms_roi_pair *jobs, int n,
                                       const vfsms_orb_params *params, int max_dist, int offset_evaluate, int32_t *out)
{
    CTX_ENTER(ctx);
    if (n < 0 || (n && (!jobs || !out)) || !params) { vfsms_set_error("attempt_orb: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    if (n == 0) return VFSMS_OK;
    TRY(ctx_prepare_orb(ctx, params));
    int c1, c2, c;
    orb_caps(params, &c1, &c2, &c);
    // a wave owns 64 queries and walks its share of the trains: split the trains so that a batch fills the chip a few times over
    const long long hwaves = (long long)((c + 63) / 64) * n;
    const int hns = (int)std::max<long long>(1, std::min<long long>(8, (8192 + hwaves - 1) / hwaves));
    size_t need = 0;
    for (int k = 0; k < n; k++) need += 2 * orb_roi_bytes(params, jobs[k].h, jobs[k].w, c1, c2, c) + match_bytes(c, hns);
    need += (sizeof(OrbDev) * 2 + sizeof(MatchDev)) * n + (64 * 2 + VFSMS_ATTEMPT_INTS) * sizeof(int) * (size_t)n + 65536;
    TRY(ctx_arena_reserve(ctx, need));
    std::vector<OrbDev> R(2 * n);
    std::vector<MatchDev> M(n);
    // counters of all ROIs and results of all jobs live in two contiguous blocks: two D2H copies per batch
    int *cblock = (int *)ctx_arena_alloc(ctx, sizeof(int) * 64 * 2 * n);
    int32_t *rblock = (int32_t *)ctx_arena_alloc(ctx, sizeof(int32_t) * VFSMS_ATTEMPT_INTS * n);
    // ROIs of one shape next to each other: the image-sized kernels are launched per shape run (launch_orb); slot s holds job ord[s]
    std::vector<int> ord(n);
    for (int k = 0; k < n; k++) ord[k] = k;
    std::stable_sort(ord.begin(), ord.end(), [&](int a_, int b_) {
        return jobs[a_].h != jobs[b_].h ? jobs[a_].h < jobs[b_].h : jobs[a_].w < jobs[b_].w; });
    for (int s_ = 0; s_ < n; s_++) {
        const int k = ord[s_];
        const uint8_t *pa, *pb; int sa, sb;
        TRY(resolve_job(ctx, jobs[k], &pa, &sa, &pb, &sb));
        TRY(orb_roi_carve(ctx, &R[2 * s_], pa, sa, jobs[k].h, jobs[k].w, params, c1, c2, c));
        TRY(orb_roi_carve(ctx, &R[2 * s_ + 1], pb, sb, jobs[k].h, jobs[k].w, params, c1, c2, c));
        for (int e = 0; e < 2; e++) {
            OrbDev &r = R[2 * s_ + e];
            r.counters = cblock + 64 * (2 * s_ + e); r.thr1 = r.counters + 16; r.n1 = r.counters + 32; r.n2 = r.counters + 48;
        }
        memset(&M[s_], 0, sizeof(MatchDev));
        TRY(match_carve(ctx, &M[s_], c, 32, hns));
        M[s_].result = rblock + VFSMS_ATTEMPT_INTS * k;
        M[s_].q = (const float *)R[2 * s_].desc; M[s_].t = (const float *)R[2 * s_ + 1].desc;
        M[s_].nq_ptr = R[2 * s_].counters + 1; M[s_].nt_ptr = R[2 * s_ + 1].counters + 1;
        M[s_].kq = R[2 * s_].kps_xy; M[s_].kt = R[2 * s_ + 1].kps_xy;
    }
    ctx->pinned_off = 0;
    OrbDev *dR; MatchDev *dM;
    TRY(upload_pinned(ctx, R.data(), sizeof(OrbDev) * 2 * n, (void **)&dR));
    TRY(upload_pinned(ctx, M.data(), sizeof(MatchDev) * n, (void **)&dM));
    TRY(launch_orb(ctx, dR, R.data(), 2 * n, params));
    TRY(launch_hamming_mode(ctx, dM, n, c, hns, max_dist, offset_evaluate));
    std::vector<int> counters((size_t)64 * 2 * n);
    HIP_TRY(hipMemcpyAsync(counters.data(), cblock, sizeof(int) * 64 * 2 * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, rblock, sizeof(int32_t) * VFSMS_ATTEMPT_INTS * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 2 * n; k++)
        if (counters[(size_t)64 * k + 2]) { vfsms_set_error("attempt_orb: internal keypoint capacity exceeded in ROI %d", k); return VFSMS_ERR_CAPACITY; }
    return VFSMS_OK;
}

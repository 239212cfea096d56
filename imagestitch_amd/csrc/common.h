// common.h -- internal declarations shared by the libvfsms translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <condition_variable>
#include <list>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/vfsms.h"

#define VFSMS_MAX_LAYERS 32      // (nOctaveLayers + 2) * nOctaves
#define VFSMS_MAX_WIN 768        // SURF descriptor window side upper bound (size <= 264 -> 739)

// ---- error plumbing ------------------------------------------------------------------------------
void vfsms_set_error(const char *fmt, ...);
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            vfsms_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return VFSMS_ERR_HIP;                                                             \
        }                                                                                     \
    } while (0)
#define TRY(expr)                  \
    do {                           \
        int _r = (expr);           \
        if (_r != VFSMS_OK) return _r; \
    } while (0)

// ---- fast-Hessian layer description (host-built, device-resident table) ---------------------------
#define VFSMS_MAX_OCTAVES 8
// resizeHaarPattern: corner c (x1, y1, x2, y2) of box k (Dx 0-2, Dy 3-5, Dxy 6-9) of the 9 x 9 pattern scaled to `size`:
// cvRound(size / 9.f * v) == (2 size v + 9) / 18 -- 2 size v is never an odd multiple of 9, so there are no ties to break
// (ctx_prepare_surf checks the table it builds with the float expression against this one).
#ifdef __HIPCC__
__host__ __device__
#endif
constexpr int vfsms_haar_corner(int size, int k, int c)
{
    constexpr int src[10][4] = { {0, 2, 3, 7}, {3, 2, 6, 7}, {6, 2, 9, 7}, {2, 0, 7, 3}, {2, 3, 7, 6}, {2, 6, 7, 9},
                                 {1, 1, 4, 4}, {5, 1, 8, 4}, {1, 5, 4, 8}, {5, 5, 8, 8} };
    return (2 * size * src[k][c] + 9) / 18;
}
// the 9 x 9 pattern coordinate (0..9) behind corner c of box k: the integral ROW of a tap depends on it alone, which is what the
// row-staged coarse Hessian (k_hessian_rows2) indexes its LDS row slots with
#ifdef __HIPCC__
__host__ __device__
#endif
constexpr int vfsms_haar_src(int k, int c)
{
    constexpr int src[10][4] = { {0, 2, 3, 7}, {3, 2, 6, 7}, {6, 2, 9, 7}, {2, 0, 7, 3}, {2, 3, 7, 6}, {2, 6, 7, 9},
                                 {1, 1, 4, 4}, {5, 1, 8, 4}, {1, 5, 4, 8}, {5, 5, 8, 8} };
    return src[k][c];
}
struct LayerPat {
    int size, step, margin, octave;   // margin = (size/2)/step
    int box[10][4];                   // dx1, dy1, dx2, dy2 for Dx[3], Dy[3], Dxy[4]  (resizeHaarPattern)
    float w[10];
};

struct SurfTables {                   // orientation lattice + descriptor Gaussian (SURFInvoker ctor)
    int nOriSamples;
    int aptx[128], apty[128];
    float aptw[128];
    float DW[400];
    // sliding-window membership of a rounded gradient angle a (0..360): bit w of the 72-bit row a is set when the ORI_WIN = 60 degree
    // window starting at 5 w holds it, i.e. |a - 5 w| < 30 or > 330 (SURFInvoker's test, evaluated once per angle on the host)
    uint32_t oriMask[361][3];
};

struct Cand {                         // NMS survivor before sorting
    float x, y, size, response;
    int octave, class_id;
    int layer, i, j;
};

// ---- one ROI's device working set; arrays of these drive every batched kernel ---------------------
#define VFSMS_PATCH_ROW 464       // 441 patch bytes, then at VFSMS_PATCH_TRIG the (sin, cos) of the window rotation
#define VFSMS_PATCH_TRIG 448
struct RoiDev {
    const uint8_t *img;
    int stride, h, w;
    int32_t *sum;                     // (h+1) x (w+1)
    uint16_t *pair;                   // h x w: pixel (y, x) in the low byte, pixel (min(y+1, h-1), x) in the high byte (k_pair_rows)
    int32_t *icarry; int ipitch;      // integral-image band carries: ceil(h/16) x ipitch column sums (ipitch = w rounded up to 4)
    float *det[VFSMS_MAX_LAYERS];
    int cap;
    int *counters;                    // [0] n candidates, [1] n kept after deletion, [2] overflow flag
    Cand *cand;
    vfsms_keypoint *kps;              // sorted (KeypointGreater), angle filled by orientation; size=-1 -> deleted
    uint8_t *patch;                   // cap x VFSMS_PATCH_ROW: 21 x 21 resized descriptor windows, rows aligned with kps
    int *keep_pos;                    // exclusive scan of keep flags
    int *order;                       // surviving keypoints grouped by descriptor-window class (largest first); counts in counters[12..15]
    float *kps_xy;                    // compacted [n][2]
    float *desc;                      // compacted [n][D]
    vfsms_keypoint *kps_out;          // compacted
};

// ---- descriptor work list of a launch (round 5) -------------------------------------------------------------
// One 32-byte record per surviving keypoint, written by k_desc_recs in the order the descriptor kernels draw them: everything a workgroup
// needs to start on a keypoint in ONE scalar load -- before, a ticket led through counters -> order[] -> kps[] -> the trig values behind the
// patch row, four dependent trips to memory with the workgroup's four waves waiting.
struct DescRec { int roi, k, win, pad; float sin_dir, cos_dir, x, y; };      // win = (int)(21 * size * 1.2f / 9) unclamped
// The plan of a launch (k_desc_plan, one workgroup): per ticket head q (= XCD) the slice of the record arrays it serves.  Head q owns the ROIs
// q, q + 8, ...; its slice of the BIG array (windows > 64 px, k_describe) is [class 0 of its ROIs, ROI after ROI][class 1 + 2 of its first
// ROI][of its second] ..., of the SMALL array (k_describe_small) [class 3 of its first ROI][of its second] ...
#define DESC_PLAN_HEADS 8
#define VFSMS_MAX_ROIS 256        // ROIs of one fused launch
struct DescPlan {
    int big_start[DESC_PLAN_HEADS], big_n0[DESC_PLAN_HEADS], big_tickets[DESC_PLAN_HEADS];     // records before the head's slice, its class-0 records, its tickets
    int small_start[DESC_PLAN_HEADS], small_tickets[DESC_PLAN_HEADS];
    int split;                        // 21: every class-0 keypoint is 21 tickets, one per output row of its patch (small batches); else 1
    int seg_base[VFSMS_MAX_ROIS][4];  // record index of the first keypoint of (roi, class) in its array
};

// ---- ORB working set of one ROI --------------------------------------------------------------------------
#define VFSMS_ORB_MAX_LEVELS 8
struct OrbDev {
    int h, w;
    uint8_t *lv[VFSMS_ORB_MAX_LEVELS]; int ls[VFSMS_ORB_MAX_LEVELS];      // pyramid level images + row strides (level 0 = the ROI itself)
    int lw[VFSMS_ORB_MAX_LEVELS], lh[VFSMS_ORB_MAX_LEVELS]; float lscale[VFSMS_ORB_MAX_LEVELS];
    uint8_t *bl[VFSMS_ORB_MAX_LEVELS];                                    // blurred levels (descriptor sampling)
    uint8_t *score[VFSMS_ORB_MAX_LEVELS]; uint8_t *nms[VFSMS_ORB_MAX_LEVELS];
    int *hist;                        // [levels][256] FAST score histogram of NMS survivors
    int *counters;                    // [1] keypoints, [2] overflow; thr1 / n1 / n2 live in the same block
    int *thr1; int *n1; int *n2;
    int cap1, cap2, cap;
    int *k1_xy; float *k1_resp;       // per level: survivors of the FAST-score cut (+ Harris response)
    int *k2_xy; float *k2_resp; float *k2_angle;   // per level: final keypoints
    float *kps_xy; uint8_t *desc; vfsms_keypoint *kps_out;                // level-major final arrays
};
struct OrbTables { int nfeat[VFSMS_ORB_MAX_LEVELS]; int umax[34]; int half_patch; int patch_size; int kf[7]; int pattern[1024]; };

// ---- one image of an enhancement batch (equalizeHist / CLAHE before detectAndDescribe) --------------------------------------
struct EnhJob { const uint8_t *src; int stride, h, w, eh, ew; uint8_t *dst; int *hist; uint8_t *lut; };   // eh, ew: size extended to the CLAHE grid

// ---- device-resident feature set (keypoints + descriptors of one image; Stitcher.tempImageFeature's payload) ---------------
struct FeatRec { float *kps_xy; void *desc; int n, dim, is_orb; int64_t block = 0; };   // block != 0: kps_xy / desc point into a shared allocation (feat_blocks)
struct FeatBlock { void *base; int refs; };

// ---- one (query ROI, train ROI) matching job --------------------------------------------------------
struct MatchDev {
    const float *q; const float *t;   // descriptors
    const int *nq_ptr; const int *nt_ptr;   // device-side counts (RoiDev.counters+1) or host-filled
    const float *kq; const float *kt; // keypoints xy (may be null for raw matching)
    int capq;
    int dim;
    // partial 2-NN per (split, query)
    float *p_d1; float *p_d2; int *p_i1; int nsplit;
    // MFMA candidate filter (fused SURF path): per (query, split, lane half) lists of (score bits, train index) + their counts
    uint2 *c_ent; int *c_cnt;
    float2 *c_m12;                    // per (list, query): best / second-best hi-only score of the bounds pass
    unsigned short *q16, *t16;        // split-bf16 operands of the filter (k_bf_split16): BF16_ROW uint16 per descriptor row
    // merged
    float *d1; float *d2; int *i1;
    int *match_flag; int *match_pos;
    int32_t *pairs;                   // [capq][2] (train, query)
    int32_t *votes;                   // [capq][2] (dx, dy) after dropping (0,0)
    int *mcount;                      // [0] n matches, [1] n votes
    int32_t *result;                  // VFSMS_ATTEMPT_INTS
    int pairs_given;                  // pairs[] supplied by the caller (vfsms_mode_offset): do not rewrite
};

// ---- context ------------------------------------------------------------------------------------------
// pending: an async upload the compute stream has not yet waited for; bytes: size of an owned allocation;
// fill: 0 filled (or being copied: `ready` is recorded), 1 reserved -- a decoder thread still owes the pixels (vfsms_tile_fill), 2 the decoder gave up
struct TileRec { uint8_t *ptr; int h, w, stride; bool owned; hipEvent_t ready; bool pending; int ch = 1; size_t bytes = 0; int fill = 0; };
struct StageBuf { uint8_t *ptr; size_t bytes; };                    // device staging of one decoded source image (vfsms_tile_fill_pair)
struct PoolEnt { size_t bytes; uint8_t *ptr; hipEvent_t idle; };   // a freed tile buffer; idle: recorded on the compute stream when the tile was freed
struct CanvasRec { uint8_t *pix; uint8_t *mask; int rows, cols, ch; int *d_err; void *scratch; std::vector<int32_t> placed; };   // d_err: sticky "degenerate fuse geometry" flag for calls made without an info readback; scratch: the fuse's statistics records + ramps; placed: (y0, x0, y1, x1) of every tile rectangle written so far = the canvas's validity (canvas_fuse_device counts the valid pixels of a ROI from it)
struct FftPlan { int M, N, nb; void *fwd, *inv, *fwd_info, *inv_info; size_t fwd_work, inv_work; };   // rocfft_plan / rocfft_execution_info
struct PhaseJobHost { const uint8_t *a, *b; int sa, sb; };
struct ProfRec { int id; hipEvent_t a, b; };

struct vfsms_ctx {
    int device;
    hipStream_t stream;
    // bump arena for per-call scratch
    char *arena; size_t arena_size; size_t arena_off;
    // pinned staging for small results
    char *pinned; size_t pinned_size; size_t pinned_off;
    int kp_cap_override;
    // SURF tables
    vfsms_surf_params cur_params; bool tables_valid;
    LayerPat *d_layers; int n_layers;
    SurfTables *d_tables;
    void *d_area_tab = nullptr;          // INTER_AREA tables of every descriptor-window size (ctx_prepare_area_tab)
    vfsms_orb_params cur_orb; bool orb_valid; OrbTables *d_orb_tables;
    std::unordered_map<int64_t, TileRec> tiles;
    std::mutex tiles_mu; std::condition_variable tiles_cv;   // reserved tiles are filled by other threads (vfsms_tile_fill*): they look tiles up and write TileRec::fill / pending under this mutex and enqueue on the copy stream; the map's structure, the buffer / event pools and the arena belong to the context's own thread
    std::mutex stage_mu; std::vector<StageBuf> stage_pool;   // device staging buffers of the decoder threads (vfsms_tile_fill_pair), the pools they share
    std::vector<StageBuf> pin_pool;                          // pinned host staging of the same threads (also under stage_mu)
    hipStream_t copy_stream;                                  // H2D uploads of tiles, overlapped with compute (vfsms_tile_upload_async)
    // second compute stream: the MFMA-bound 2-NN search of the first part of a batch runs on it beside the VALU / TA-bound detect stage of
    // the second part (attempt_surf_impl); created on first use, always joined back into `stream` before a call returns
    hipStream_t stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<PoolEnt> tile_pool;      // freed tile buffers, reused by allocation size (no hipMalloc / hipFree per step)
    size_t tile_pool_bytes = 0;
    std::vector<hipEvent_t> event_pool;
    std::unordered_map<int64_t, CanvasRec> canvases;
    CanvasRec spare_canvas; bool has_spare_canvas = false;   // the buffers of the last canvas freed: a session's mosaics are of one size, and hipMalloc / hipFree of a canvas (28 GB at configs[4]) cost more than the walk
    std::unordered_map<int64_t, FeatRec> feats;
    std::unordered_map<int64_t, FeatBlock> feat_blocks;      // one allocation for the sets of a batch (vfsms_features_surf_batch), freed with its last set
    int64_t next_handle;
    std::list<FftPlan> plans;            // list: get_plan hands out stable pointers
    std::vector<std::pair<int, void *>> fft_tabs;   // twiddle tables exp(-2 pi i q / L) of the LDS transforms of phase_kernels.hip, one per length
    // optional per-stage timing with HIP events on this context's stream (vfsms_profile_*)
    bool prof_on;
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    std::vector<std::string> prof_names;
    std::vector<double> prof_ms;
    std::vector<long long> prof_calls;
};

int prof_begin(vfsms_ctx *ctx, const char *name);      // returns a record index or -1 when profiling is off
void prof_end(vfsms_ctx *ctx, int rec);
struct ProfScope {
    vfsms_ctx *c; int r;
    ProfScope(vfsms_ctx *ctx, const char *name) : c(ctx), r(prof_begin(ctx, name)) {}
    ~ProfScope() { prof_end(c, r); }
};

int ctx_arena_reserve(vfsms_ctx *ctx, size_t bytes);             // ensure capacity (may sync + realloc), reset offset
void *ctx_arena_alloc(vfsms_ctx *ctx, size_t bytes, size_t align = 256);
int ctx_prepare_surf(vfsms_ctx *ctx, const vfsms_surf_params *p);

// ---- kernel launchers (each is stream-ordered, no host sync) --------------------------------------------
// surf_kernels.hip
size_t surf_roi_bytes(int h, int w, int cap, int nlayers_total, int noctaves, int dim);
int surf_roi_carve(vfsms_ctx *ctx, RoiDev *r, const uint8_t *img, int stride, int h, int w, int cap,
                   const vfsms_surf_params *p);
int launch_integral(vfsms_ctx *ctx, const RoiDev *d_rois, int nrois, int maxh, int maxw);
size_t integral_carry_bytes(int h, int w);
int launch_surf_detect(vfsms_ctx *ctx, const RoiDev *d_rois, const RoiDev *h_rois, int nrois,
                       const vfsms_surf_params *p);
int launch_surf_describe(vfsms_ctx *ctx, const RoiDev *d_rois, const RoiDev *h_rois, int nrois,
                         const vfsms_surf_params *p);
// match_kernels.hip
size_t match_bytes(int capq, int nsplit);
int match_carve(vfsms_ctx *ctx, MatchDev *m, int capq, int dim, int nsplit);
int launch_max_norm2_d64(vfsms_ctx *ctx, const float *a, int n, unsigned *d_out);
size_t match_filter_bytes(int capq, int capt, int cns);
int match_filter_carve(vfsms_ctx *ctx, MatchDev *m, int capq, int capt, int cns);
int launch_bf_l2_filtered(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, int capt, int cns);
int launch_bf_l2(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, int nsplit, int dim);
int launch_merge_only(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq);
int launch_ratio_only(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, double ratio);
int launch_ratio_mode(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, double ratio, int offset_evaluate);
int launch_mode_only(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capm, int offset_evaluate);
int launch_bf_hamming(vfsms_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes,
                      int *best_idx, int *best_dist);
int launch_scan_mode(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, int offset_evaluate);
// orb_kernels.hip
int ctx_prepare_orb(vfsms_ctx *ctx, const vfsms_orb_params *p);
size_t orb_roi_bytes(const vfsms_orb_params *p, int h, int w, int cap1, int cap2, int cap);
int orb_roi_carve(vfsms_ctx *ctx, OrbDev *r, const uint8_t *img, int stride, int h, int w, const vfsms_orb_params *p,
                  int cap1, int cap2, int cap);
int launch_orb(vfsms_ctx *ctx, const OrbDev *d_rois, const OrbDev *h_rois, int nrois, const vfsms_orb_params *p);
int launch_hamming_mode(vfsms_ctx *ctx, const MatchDev *d_jobs, int njobs, int capq, int nsplit, int max_dist, int offset_evaluate);
// phase_kernels.hip
int phase_correlate_device(vfsms_ctx *ctx, const uint8_t *a, int stride_a, const uint8_t *b, int stride_b,
                           int h, int w, double *d_out3);
int phase_correlate_batch_device(vfsms_ctx *ctx, const PhaseJobHost *jobs, int nb, int h, int w, double *d_out3);
int phase_bytes(vfsms_ctx *ctx, int h, int w, int nb, size_t *bytes);
int ctx_upload_small(vfsms_ctx *ctx, const void *src, size_t bytes, void **d);   // launch records through the pinned staging buffer
// enhance_kernels.hip
size_t enhance_scratch_bytes(int h, int w, int mode, int tiles);
int enhance_carve(vfsms_ctx *ctx, EnhJob *J, const uint8_t *src, int stride, int h, int w, int mode, int tiles);
int launch_enhance(vfsms_ctx *ctx, const EnhJob *d_jobs, const EnhJob *h_jobs, int n, int mode, double clip_limit, int tiles);
// fuse_kernels.hip
int canvas_fuse_device(vfsms_ctx *ctx, CanvasRec *cv, const uint8_t *d_tile, int h, int w, int y0, int x0,
                       int ry0, int rx0, int ry1, int rx1, int dx, int dy, int32_t *info, int method = 0);   // method 0 fadeInAndFadeOut, 1 trigonometric
int canvas_blend_device(vfsms_ctx *ctx, CanvasRec *cv, const uint8_t *d_tile, int h, int w, int y0, int x0,
                        int ry0, int rx0, int ry1, int rx1, int mode);
int canvas_paste_device(vfsms_ctx *ctx, CanvasRec *cv, const uint8_t *d_tile, int h, int w, int y0, int x0);
size_t canvas_scratch_bytes(int rows, int cols);
int canvas_scratch_init(vfsms_ctx *ctx, CanvasRec *cv);

#ifdef __HIPCC__
// Speed only (placement is not a contract): workgroups are observed to land on XCD (linear block id % 8), each XCD with a private
// 4 MB L2.  Gather-heavy kernels whose slowest grid index is a unit with its own working set (an ROI and its 3.4 MB integral
// image; a (match job, train chunk) and its descriptors) hand every XCD WHOLE units, so that a unit's data is pulled into one L2
// once instead of into all eight.  L = linear block id, per_unit blocks per unit; the units beyond the last multiple of 8 keep
// the plain order.
__device__ __forceinline__ void xcd_roi_map(unsigned L, unsigned per_unit, unsigned nunits, unsigned &unit, unsigned &inner)
{
    const unsigned nfull = nunits & ~7u;
    if (L < per_unit * nfull) { const unsigned x = L & 7u, slot = L >> 3; unit = x + 8u * (slot / per_unit); inner = slot % per_unit; }
    else { unit = L / per_unit; inner = L % per_unit; }
}
#endif

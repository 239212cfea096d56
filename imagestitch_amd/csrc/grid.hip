// grid.hip -- whole-path registration behind ONE entry point: the candidate state machine of the reference's incremental search
// (Stitcher.calculateOffsetForFeatureSearchIncre / ...PhaseCorrleateIncre, Stitcher.py:306-367 / 205-258) driven over every
// consecutive pair of a shooting path (the loop of Stitcher.flowStitch, Stitcher.py:64-79) with speculative fused batches.
//
// Host-side C++ (no kernels here): it decides WHICH (pair, direction, i) attempts go into the next fused device batch and selects
// results in the reference's candidate order, exactly like imagestitch_amd/grid.py:GridRegistrar.chain, whose behaviour it
// reproduces decision for decision (tests/test_grid_registrar.py runs both against the sequential search on random truth tables
// through vfsms_pairs_offsets_eval, which takes the attempt evaluator as a callback and needs no GPU).
//
//   * an attempt is a pure function of (pair, direction, i): what is evaluated together is free, the accepted candidate is always
//     the first one in the order  for i in 1..maxI-1 { d = d0; do { (d, i); d = rotate(d) } while (d != d0) }   (Stitcher.py:319-351);
//   * `self.direction` threads from pair to pair (Stitcher.py:252,361): a successful pair hands its direction on, a failed pair
//     leaves it untouched;
//   * prediction (run-length history of the serpentine, ring position that resolved the last turn, plan-ahead over the predicted
//     path) only bounds the speculation window: it changes how many attempts are wasted, never a result.
#include "common.h"
#include <math.h>
#include <string.h>
#include <map>
#include <set>
#include <tuple>
#include <vector>
#include <algorithm>

namespace {

typedef std::tuple<int, int, int> Key;          // (pair, direction, i)
struct Attempt { bool ok; int a, b, v; };      // status, raw dx, raw dy, votes
struct Row { int v[6]; };                       // status, dx, dy, direction, i, votes

inline int rotate_dir(int d, int incre)
{
    d += incre;
    if (d == 5) d = 1;
    if (d == 0) d = 4;
    return d;
}

struct Chain {
    vfsms_attempt_eval eval; void *user;
    int (*ready)(void *user, int tile) = nullptr;         // optional: is this tile's image on the device yet? (tiles filled by decoder threads)
    const int32_t *shapes; int n_tiles;
    const vfsms_grid_params *P;
    std::map<Key, Attempt> cache;
    std::map<std::pair<int, int>, std::pair<Row, int>> memo;   // (pair, incoming direction) -> (row, next direction): chains that start blind share it
    long long n_attempts = 0, n_batches = 0;

    int maxI() const { return (int)(floor(0.5 / P->roi_ratio) + 1) + 1; }

    std::vector<std::pair<int, int>> ring(int d0, int i) const
    {
        std::vector<std::pair<int, int>> r;
        int d = d0;
        do { r.push_back(std::make_pair(d, i)); d = rotate_dir(d, P->direct_incre); } while (d != d0);
        return r;
    }

    // evaluate whatever of `items` is not cached yet (first occurrence order, pairs < last only) as ONE batch
    int evaluate(const std::vector<Key> &items, int last)
    {
        std::vector<vfsms_attempt_key> todo;
        std::set<Key> seen;
        for (const Key &it : items) {
            if (std::get<0>(it) >= last || cache.count(it) || seen.count(it)) continue;
            seen.insert(it);
            vfsms_attempt_key k; k.pair = std::get<0>(it); k.direction = std::get<1>(it); k.i = std::get<2>(it);
            todo.push_back(k);
        }
        if (todo.empty()) return VFSMS_OK;
        if (ready) {
            // Ingest pipeline: a batch takes the attempts of its first pair (the one the chain is waiting for -- the evaluator blocks until
            // those two tiles are there) and, of the speculative rest, only what is decoded already: the batches grow with the decoder
            // pool's progress instead of making the first one wait for a whole window of tiles.  What is left out is simply not cached.
            size_t keep = 0;
            for (size_t n = 0; n < todo.size(); n++)
                if (todo[n].pair == todo[0].pair || (ready(user, todo[n].pair) && ready(user, todo[n].pair + 1))) todo[keep++] = todo[n];
            todo.resize(keep);
        }
        std::vector<int32_t> rows((size_t)todo.size() * VFSMS_ATTEMPT_INTS, 0);
        const int rc = eval(user, todo.data(), (int)todo.size(), rows.data());
        if (rc != VFSMS_OK) return rc;
        n_attempts += (long long)todo.size(); n_batches += 1;
        for (size_t n = 0; n < todo.size(); n++) {
            const int32_t *r = &rows[n * VFSMS_ATTEMPT_INTS];
            Attempt a;
            if (P->method == 2) { a.ok = r[0] != 0; a.a = r[1]; a.b = r[2]; a.v = 0; }                 // phase: evaluator applied the response gate
            else { a.ok = r[0] != 0 && r[4] > 0 && r[5] > 0; a.a = r[1]; a.b = r[2]; a.v = r[3]; }   // features: an image without keypoints never matches
            cache[Key(todo[n].pair, todo[n].direction, todo[n].i)] = a;
        }
        return VFSMS_OK;
    }

    // Stitcher.py:352-360: ROI-relative vote -> full-tile offset
    void correct(int &dx, int &dy, int d, int i, int k) const
    {
        const int ah = shapes[2 * k], aw = shapes[2 * k + 1], bh = shapes[2 * (k + 1)], bw = shapes[2 * (k + 1) + 1];
        const double f = (double)i * P->roi_ratio;
        if (d == 1) dx = dx + ah - (int)(f * ah);
        else if (d == 2) dy = dy + aw - (int)(f * aw);
        else if (d == 3) dx = dx - (bh - (int)(f * bh));
        else if (d == 4) dy = dy - (bw - (int)(f * bw));
    }

    int run(int first, int last, int d_in, int midpath, int stop_on_fail, int32_t *out, int32_t *d_out)
    {
        const int window = std::max(1, (int)P->window);
        std::vector<int> runs;
        int run_len = 0, slow = 1, prev_d = 0, d = d_in;
        std::map<int, int> ring_hint;                        // direction -> ring position that resolved the last turn from it
        std::map<std::pair<int, int>, int> trans2;           // (direction before, direction) -> direction the next turn led to
        if (P->path_hint && P->path_hint_len > 0 && first > 0) {
            // A chain that starts inside the path: the predictor is primed with the history the PREDICTED directions imply for the pairs
            // before `first` (run lengths, the direction each turn led to, the ring position that resolved it), as if this chain had
            // registered them itself.  Bookkeeping only -- nothing is evaluated, nothing cached.
            int hd = P->path_hint[0];
            const int upto = std::min(first, (int)P->path_hint_len);
            for (int k = 0; k < upto && hd >= 1 && hd <= 4; k++) {
                const int nd = P->path_hint[k];
                if (nd < 1 || nd > 4) break;
                if (nd == hd) { run_len += 1; slow = std::min(2 * slow, window); }
                else {
                    runs.push_back(run_len);
                    run_len = 1; slow = 1;
                    trans2[std::make_pair(prev_d, hd)] = nd;
                    const std::vector<std::pair<int, int>> rg = ring(hd, 1);
                    for (size_t q = 0; q < rg.size(); q++) if (rg[q].first == nd) { ring_hint[hd] = (int)q; break; }
                    prev_d = hd;
                }
                hd = nd;
            }
            if (hd != d_in) { runs.clear(); run_len = 0; slow = 1; prev_d = 0; ring_hint.clear(); trans2.clear(); }   // entered differently than predicted: no basis
        }
        for (int k = first; k < last; k++) {
          Row row; int d_next;
          const auto mem = memo.find(std::make_pair(k, d));
          if (mem != memo.end()) { row = mem->second.first; d_next = mem->second.second; }    // another chain has been here: same state, same future
          else {
            // ---- predicted continuation of the path as one batch (up to `window` attempts)
            if (!cache.count(Key(k, d, 1))) {
                std::vector<Key> items;
                if (P->path_hint && P->path_hint_len > 0) {
                    // the predicted directions as the plan itself: the run at the current direction up to the predicted change, the candidate
                    // ring of that pair up to the predicted new direction, the next run, ... (grid.py: plan_hint)
                    int cd = d;
                    for (int kk = k; kk < last && kk < (int)P->path_hint_len && (int)items.size() < window; kk++) {
                        const int hd = P->path_hint[kk];
                        if (hd < 1 || hd > 4) break;
                        const std::vector<std::pair<int, int>> rg = ring(cd, 1);
                        int at = -1;
                        for (size_t q = 0; q < rg.size(); q++) if (rg[q].first == hd) { at = (int)q; break; }
                        if (hd == cd || at < 0) items.push_back(Key(kk, cd, 1));
                        else {
                            for (int q = 0; q <= at; q++) items.push_back(Key(kk, rg[q].first, rg[q].second));
                            cd = hd;
                        }
                    }
                }
                const bool hinted = !items.empty();           // else: the plan from this chain's own history
                std::vector<int> R(runs);
                int rl = run_len, cd = d, cp = prev_d, kk = k;
                while (!hinted && kk < last && (int)items.size() < window) {
                    if (R.size() < 2) break;
                    const int pred = R[R.size() - 2];
                    const int remaining = pred - rl;
                    if (remaining < 0) break;                 // this run already outlived the prediction: slow start instead
                    if (remaining >= 1) {
                        const int n = std::min(std::min(remaining, window - (int)items.size()), last - kk);
                        for (int t = 0; t < n; t++) items.push_back(Key(kk + t, cd, 1));
                        kk += n; rl += n;
                        if (n < remaining) break;
                        continue;
                    }
                    const std::vector<std::pair<int, int>> rg = ring(cd, 1);
                    const auto it = trans2.find(std::make_pair(cp, cd));
                    int upto = -1;
                    if (it != trans2.end())
                        for (size_t q = 0; q < rg.size(); q++) if (rg[q].first == it->second) { upto = (int)q; break; }
                    if (upto < 0) {
                        const auto h = ring_hint.find(cd);
                        const int stop = h != ring_hint.end() ? h->second : (int)rg.size() - 1;
                        for (int q = 0; q <= stop && q < (int)rg.size(); q++) items.push_back(Key(kk, rg[q].first, rg[q].second));
                        break;
                    }
                    for (int q = 0; q <= upto; q++) items.push_back(Key(kk, rg[q].first, rg[q].second));
                    R.push_back(rl); rl = 1;
                    cp = cd; cd = it->second;
                    kk += 1;
                }
                if (items.empty())                            // no history yet: slow start
                    for (int kk2 = k; kk2 < std::min(k + slow, last); kk2++)
                        if (!memo.count(std::make_pair(kk2, d))) items.push_back(Key(kk2, d, 1));
                TRY(evaluate(items, last));
                if (!cache.count(Key(k, d, 1))) { std::vector<Key> one(1, Key(k, d, 1)); TRY(evaluate(one, last)); }
            }
            // ---- select in the reference's candidate order
            bool found = false;
            int fd = 0, fi = 0, fa = 0, fb = 0, fv = 0;
            const int mI = maxI();
            for (int i = 1; i < mI && !found; i++) {
                const std::vector<std::pair<int, int>> rg = ring(d, i);
                const int ri = i - 1;
                for (int pos = 0; pos < (int)rg.size() && !found; pos++) {
                    const Key key(k, rg[pos].first, rg[pos].second);
                    if (!cache.count(key)) {
                        int h = (int)rg.size() - 1;
                        if (ri == 0) { const auto hh = ring_hint.find(d); if (hh != ring_hint.end()) h = hh->second; }
                        const int stop = std::max(pos, std::min(h, (int)rg.size() - 1));
                        std::vector<Key> items;
                        for (int q = pos; q <= stop; q++) items.push_back(Key(k, rg[q].first, rg[q].second));
                        TRY(evaluate(items, last));
                    }
                    const Attempt &a = cache[key];
                    if (a.ok) {
                        found = true; fd = rg[pos].first; fi = rg[pos].second; fa = a.a; fb = a.b; fv = a.v;
                        if (ri == 0) ring_hint[d] = pos;
                    }
                }
            }
            if (found) {
                int dx = fa, dy = fb;
                correct(dx, dy, fd, fi, k);
                row.v[0] = 1; row.v[1] = dx; row.v[2] = dy; row.v[3] = fd; row.v[4] = fi; row.v[5] = fv;
                d_next = fd;                                   // self.direction = localDirection
            } else {
                row.v[0] = 0; row.v[1] = 0; row.v[2] = 0; row.v[3] = d; row.v[4] = 0; row.v[5] = 0;
                d_next = d;                                    // a failed pair leaves self.direction untouched
            }
            memo[std::make_pair(k, d)] = std::make_pair(row, d_next);
          }
            // ---- predictor bookkeeping
            if (row.v[0] && d_next == d) {
                run_len += 1;
                slow = std::min(2 * slow, (midpath && runs.size() < 2) ? 4 : window);
            } else if (row.v[0]) {
                runs.push_back(run_len);
                run_len = 1; slow = 1;
                trans2[std::make_pair(prev_d, d)] = d_next;
                prev_d = d;
            }
            for (int c = 0; c < 6; c++) out[(size_t)(k - first) * 6 + c] = row.v[c];
            d = d_next;
            if (stop_on_fail && !row.v[0]) break;              // flowStitch discards everything behind the first break (Stitcher.py:74-76)
        }
        *d_out = d;
        return VFSMS_OK;
    }
};

// ---- the device evaluator: ROI rectangles of Method.getROIRegionForIncreMethod (ImageUtility.py:66-101) + one fused batch -------------
struct DeviceEval {
    vfsms_ctx *ctx; const int64_t *tiles; const int32_t *shapes; const vfsms_grid_params *P;
    int kp_seen = 0, kp_cap = 0;          // SURF keypoint capacity follows the largest ROI seen (capacity-sized grids, no host sync in a batch)
    long long cap_retries = 0, sum_nq_nt = 0, sum_nq_plus_nt = 0, roi_px = 0;   // workload figures for the roofline report
};

void roi_rect(int row, int col, int direction, bool first, double ratio, int *y0, int *x0, int *h, int *w)
{
    if (direction == 1 || direction == 3) {
        const int n = (int)floor(row * ratio);
        const bool at_end = (direction == 1) == first;
        *y0 = at_end ? row - n : 0; *x0 = 0; *h = n; *w = col;
    } else {
        const int n = (int)floor(col * ratio);
        const bool at_end = (direction == 2) == first;
        *y0 = 0; *x0 = at_end ? col - n : 0; *h = row; *w = n;
    }
}

int device_eval(void *user, const vfsms_attempt_key *items, int n, int32_t *rows)
{
    DeviceEval *E = (DeviceEval *)user;
    std::vector<vfsms_roi_pair> jobs(n);
    for (int k = 0; k < n; k++) {
        const int p = items[k].pair;
        const double ratio = (double)items[k].i * E->P->roi_ratio;      // searchRatio = i * roiRatio in float64 first (3 * 0.2 = 0.6000000000000001)
        int ay, ax, ah, aw, by, bx, bh, bw;
        roi_rect(E->shapes[2 * p], E->shapes[2 * p + 1], items[k].direction, true, ratio, &ay, &ax, &ah, &aw);
        roi_rect(E->shapes[2 * (p + 1)], E->shapes[2 * (p + 1) + 1], items[k].direction, false, ratio, &by, &bx, &bh, &bw);
        if (ah != bh || aw != bw) { vfsms_set_error("pairs_offsets: tiles of different size in one pair"); return VFSMS_ERR_UNSUPPORTED; }
        jobs[k].tile_a = E->tiles[p]; jobs[k].tile_b = E->tiles[p + 1];
        jobs[k].ay0 = ay; jobs[k].ax0 = ax; jobs[k].by0 = by; jobs[k].bx0 = bx; jobs[k].h = ah; jobs[k].w = aw;
        E->roi_px += 2LL * ah * aw;
    }
    const vfsms_grid_params *P = E->P;
    if (P->method == 0) {
        // adaptive capacity: 1.5 x the largest keypoint count seen + 1024; an overflow falls back to the library default and repeats
        const int user_cap = E->ctx->kp_cap_override;
        if (!user_cap && E->kp_cap) E->ctx->kp_cap_override = E->kp_cap;
        int rc = vfsms_attempt_surf_batch_enhanced(E->ctx, jobs.data(), n, &P->surf, P->search_ratio, P->offset_evaluate, P->enhance_mode,
                                                   P->clip_limit, P->tile_grid, rows);
        if (rc == VFSMS_ERR_CAPACITY && !user_cap && E->kp_cap) {
            E->kp_cap = 0; E->kp_seen = 0; E->cap_retries++;
            E->ctx->kp_cap_override = 0;
            rc = vfsms_attempt_surf_batch_enhanced(E->ctx, jobs.data(), n, &P->surf, P->search_ratio, P->offset_evaluate, P->enhance_mode,
                                                   P->clip_limit, P->tile_grid, rows);
        }
        E->ctx->kp_cap_override = user_cap;
        if (rc != VFSMS_OK) return rc;
        for (int k = 0; k < n; k++) {
            const long long nq = rows[(size_t)k * VFSMS_ATTEMPT_INTS + 4], nt = rows[(size_t)k * VFSMS_ATTEMPT_INTS + 5];
            E->sum_nq_nt += nq * nt; E->sum_nq_plus_nt += nq + nt;
        }
        for (int k = 0; k < n; k++) E->kp_seen = std::max(E->kp_seen, std::max(rows[(size_t)k * VFSMS_ATTEMPT_INTS + 4], rows[(size_t)k * VFSMS_ATTEMPT_INTS + 5]));
        const int want = (int)(E->kp_seen * 1.5) + 1024;
        if (E->kp_seen > 0 && want != E->kp_cap && (E->kp_cap == 0 || want > E->kp_cap || want < E->kp_cap * 0.6)) E->kp_cap = want;
        return VFSMS_OK;
    }
    if (P->method == 1)
        return vfsms_attempt_orb_batch(E->ctx, jobs.data(), n, &P->orb, P->orb_max_dist, P->offset_evaluate, rows);
    if (P->method == 2) {
        std::vector<double> ph((size_t)3 * n);
        TRY(vfsms_attempt_phase_batch(E->ctx, jobs.data(), n, ph.data()));
        for (int k = 0; k < n; k++) {
            // offset = [int(y), int(x)] (truncation); accepted when response > threshold (Stitcher.py:231-236)
            int32_t *r = rows + (size_t)k * VFSMS_ATTEMPT_INTS;
            r[0] = ph[3 * k + 2] > P->phase_threshold; r[1] = (int32_t)ph[3 * k + 1]; r[2] = (int32_t)ph[3 * k]; r[3] = 0; r[4] = 1; r[5] = 1;
        }
        return VFSMS_OK;
    }
    vfsms_set_error("pairs_offsets: method must be 0 (surf), 1 (orb) or 2 (phase)");
    return VFSMS_ERR_BAD_ARG;
}

int check_args(const int32_t *shapes, int n_tiles, int first, int last, int direction_in, const vfsms_grid_params *p, const int32_t *out, const int32_t *d_out)
{
    if (!shapes || !p || !out || !d_out || n_tiles < 1 || first < 0 || last < first || last > n_tiles - 1 || direction_in < 1 || direction_in > 4 ||
        p->roi_ratio <= 0 || p->roi_ratio > 0.5 || p->direct_incre < -1 || p->direct_incre > 1) {
        vfsms_set_error("pairs_offsets: bad arguments");
        return VFSMS_ERR_BAD_ARG;
    }
    return VFSMS_OK;
}

}  // namespace

// the library's own evaluator hands its readiness query to the chain it is about to start (the public *_eval entry points keep their signature)
static thread_local int (*g_ready_for_next_chain)(void *user, int tile) = nullptr;
int tile_is_filled(vfsms_ctx *ctx, int64_t handle);          // api.hip
static int device_ready(void *user, int tile)
{
    DeviceEval *E = (DeviceEval *)user;
    return tile_is_filled(E->ctx, E->tiles[tile]);
}

extern "C" int vfsms_pairs_offsets_eval(vfsms_attempt_eval eval, void *user, const int32_t *shapes_hw, int n_tiles, int first_pair, int last_pair,
                                        int direction_in, int midpath, int stop_on_fail, const vfsms_grid_params *p, int32_t *out,
                                        int32_t *direction_out, int64_t *stats)
{
    int (*const ready)(void *, int) = g_ready_for_next_chain;
    g_ready_for_next_chain = nullptr;                        // consumed whatever happens below
    if (!eval) { vfsms_set_error("pairs_offsets: null evaluator"); return VFSMS_ERR_BAD_ARG; }
    TRY(check_args(shapes_hw, n_tiles, first_pair, last_pair, direction_in, p, out, direction_out));
    memset(out, 0, sizeof(int32_t) * 6 * (size_t)(last_pair - first_pair));
    Chain C; C.eval = eval; C.user = user; C.shapes = shapes_hw; C.n_tiles = n_tiles; C.P = p;
    C.ready = ready;
    const int rc = C.run(first_pair, last_pair, direction_in, midpath, stop_on_fail, out, direction_out);
    if (stats) { for (int k = 0; k < 8; k++) stats[k] = 0; stats[0] = C.n_attempts; stats[1] = C.n_batches; }
    return rc;
}

// A chunk in the MIDDLE of a path (rank > 0 of the pair-sharded form): the direction it is entered with is the result of the
// pairs before it, which another GPU is still working on.  The chunk is therefore registered for every possible incoming direction:
// the four first candidates of its first pair go out as ONE batch, the four chains share one attempt cache and one (pair, direction)
// memo, and they merge as soon as they agree on a direction -- after the first pair, in practice.  out: [4][per][6] rows, chain d at
// out + (d - 1) * per * 6; direction_out[4].
extern "C" int vfsms_pairs_offsets_blind_eval(vfsms_attempt_eval eval, void *user, const int32_t *shapes_hw, int n_tiles, int first_pair, int last_pair,
                                              int per, const vfsms_grid_params *p, int32_t *out, int32_t *direction_out, int64_t *stats)
{
    int (*const ready)(void *, int) = g_ready_for_next_chain;
    g_ready_for_next_chain = nullptr;
    if (!eval) { vfsms_set_error("pairs_offsets: null evaluator"); return VFSMS_ERR_BAD_ARG; }
    TRY(check_args(shapes_hw, n_tiles, first_pair, last_pair, 1, p, out, direction_out));
    if (per < last_pair - first_pair) { vfsms_set_error("pairs_offsets_blind: per < chunk length"); return VFSMS_ERR_BAD_ARG; }
    memset(out, 0, sizeof(int32_t) * 6 * 4 * (size_t)per);
    Chain C; C.eval = eval; C.user = user; C.shapes = shapes_hw; C.n_tiles = n_tiles; C.P = p;
    C.ready = ready;
    if (last_pair > first_pair) {
        std::vector<Key> heads;
        for (int d = 1; d <= 4; d++) heads.push_back(Key(first_pair, d, 1));
        TRY(C.evaluate(heads, last_pair));
    }
    for (int d = 1; d <= 4; d++) {
        direction_out[d - 1] = d;
        if (last_pair > first_pair) TRY(C.run(first_pair, last_pair, d, 1, 0, out + (size_t)(d - 1) * per * 6, &direction_out[d - 1]));
    }
    if (stats) { for (int k = 0; k < 8; k++) stats[k] = 0; stats[0] = C.n_attempts; stats[1] = C.n_batches; }
    return VFSMS_OK;
}

extern "C" int vfsms_pairs_offsets_blind(vfsms_ctx *ctx, const int64_t *tiles, const int32_t *shapes_hw, int n_tiles, int first_pair, int last_pair,
                                         int per, const vfsms_grid_params *p, int32_t *out, int32_t *direction_out, int64_t *stats)
{
    if (!ctx || !tiles) { vfsms_set_error("pairs_offsets: null context / tiles"); return VFSMS_ERR_BAD_ARG; }
    DeviceEval E; E.ctx = ctx; E.tiles = tiles; E.shapes = shapes_hw; E.P = p;
    g_ready_for_next_chain = device_ready;
    const int rc = vfsms_pairs_offsets_blind_eval(device_eval, &E, shapes_hw, n_tiles, first_pair, last_pair, per, p, out, direction_out, stats);
    if (stats) { stats[2] = E.cap_retries; stats[3] = E.sum_nq_nt; stats[4] = E.sum_nq_plus_nt; stats[5] = E.roi_px; }
    return rc;
}

extern "C" int vfsms_pairs_offsets(vfsms_ctx *ctx, const int64_t *tiles, const int32_t *shapes_hw, int n_tiles, int first_pair, int last_pair,
                                   int direction_in, int midpath, int stop_on_fail, const vfsms_grid_params *p, int32_t *out,
                                   int32_t *direction_out, int64_t *stats)
{
    if (!ctx || !tiles) { vfsms_set_error("pairs_offsets: null context / tiles"); return VFSMS_ERR_BAD_ARG; }
    DeviceEval E; E.ctx = ctx; E.tiles = tiles; E.shapes = shapes_hw; E.P = p;
    g_ready_for_next_chain = device_ready;
    const int rc = vfsms_pairs_offsets_eval(device_eval, &E, shapes_hw, n_tiles, first_pair, last_pair, direction_in, midpath, stop_on_fail, p, out,
                                            direction_out, stats);
    if (stats) { stats[2] = E.cap_retries; stats[3] = E.sum_nq_nt; stats[4] = E.sum_nq_plus_nt; stats[5] = E.roi_px; }
    return rc;
}

"""Batched, device-resident registration of a whole shooting path, and its pair-sharded multi-GPU form.

The reference registers pairs one at a time (Stitcher.flowStitch, Stitcher.py:64-79) and threads one piece of
state from pair to pair: `self.direction` (Stitcher.py:252,361), which decides the ORDER in which the
(direction, ROI growth i) candidates of the next pair are tried -- and with offsetEvaluate = 3 the first
accepted candidate wins, so the order is observable.  GridRegistrar produces exactly the sequential result
while keeping the GPU full:

  * speculation: a window of consecutive pairs is evaluated in ONE fused batch (SURF + BF-L2 + ratio + mode,
    or phase correlation) under the assumption that the direction does not change; the first pair whose
    (direction, i = 1) attempt fails is then resolved candidate ring by candidate ring, and speculative results
    behind it are kept only if the direction it ends in is the one they assumed;
  * sharding (one process per GPU): ranks take contiguous chunks of the path.  A rank other than 0 does not
    know the direction its first pair inherits, so it follows the chain for every possible incoming direction
    (chains merge as soon as they agree, normally after one pair) and ONE all-gather of int32 offset tables
    (RCCL over xGMI when the process group is "nccl") lets every rank select the consistent chain.
"""
import numpy as np

from .utility import roi_rect

RESULT_INTS = 6   # status, dx, dy, direction, i, votes


def _rotate(direction, incre):
    direction += incre
    if direction == 5:
        direction = 1
    if direction == 0:
        direction = 4
    return direction


class GridRegistrar:
    def __init__(self, engine, method="surf", roiRatio=0.2, searchRatio=0.75, offsetEvaluate=3, directIncre=1,
                 surfParams=None, phaseResponseThreshold=0.15, window=16, enhance=(0, 0.0, 0)):
        self.eng = engine
        self.method = method
        self.roiRatio = roiRatio
        self.searchRatio = searchRatio
        self.offsetEvaluate = offsetEvaluate
        self.directIncre = directIncre
        self.params = surfParams
        self.phaseThr = phaseResponseThreshold
        self.window = max(1, int(window))
        self.enhance = tuple(enhance)                     # (mode, clipLimit, tileSize) of Method.isEnhance (Stitcher.py:327-334)
        self.stats = dict(attempts=0, batches=0, sum_nq_nt=0, sum_nq_plus_nt=0, sum_nq=0, roi_px=0)
        # Path memory: the accepted directions of the last path this registrar registered are the PREDICTION for the next one of the same
        # length (a session shoots one scan pattern after the other: Main.py loops over its datasets with one setting) -- they drive the
        # speculation plan from the first pair on and, in the sharded form, the chunk starts.  A prior like a branch predictor's: every
        # attempt is still evaluated, results never depend on it (tests/test_grid_registrar.py).
        self.remember = True
        self.path_memory = None
        # A memory that mispredicts is not kept on trust (_learn): `path_suspect` marks a memory adopted from a path the previous memory had
        # mispredicted; a second misprediction in a row drops the memory and the next path is registered cold.
        self.path_suspect = False
        self.mispredictions = 0

    # -- candidate order of Stitcher.py:319-351 --------------------------------------------------------------
    def maxI(self):
        return int(np.floor(0.5 / self.roiRatio) + 1) + 1

    def rings(self, d0):
        """[[(direction, i), ...] per i]: each i restarts at d0 and rotates until it is back at d0."""
        out = []
        for i in range(1, self.maxI()):
            ring, d = [], d0
            while True:
                ring.append((d, i))
                d = _rotate(d, self.directIncre)
                if d == d0:
                    break
            out.append(ring)
        return out

    # -- one batch of attempts -----------------------------------------------------------------------------------
    def _attempts(self, handles, shapes, items):
        """items: [(pair index k, direction, i)] -> [(status, raw_dx, raw_dy, votes)]"""
        jobs = []
        for (k, d, i) in items:
            ra = roi_rect(shapes[k], d, "first", i * self.roiRatio)
            rb = roi_rect(shapes[k + 1], d, "second", i * self.roiRatio)
            if ra[2:] != rb[2:]:
                raise ValueError("tiles of different size in one pair are not supported by the batched path")
            jobs.append((handles[k], handles[k + 1], ra[0], ra[1], rb[0], rb[1], ra[2], ra[3]))
            self.stats["roi_px"] += 2 * ra[2] * ra[3]
        self.stats["attempts"] += len(jobs)
        self.stats["batches"] += 1
        if self.method == "surf":
            rows = self._surf_batch(jobs)
            nq = rows[:, 4].astype(np.int64); nt = rows[:, 5].astype(np.int64)
            self.stats["sum_nq_nt"] += int((nq * nt).sum())
            self.stats["sum_nq_plus_nt"] += int((nq + nt).sum())
            self.stats["sum_nq"] += int(nq.sum())
            return [(bool(r[0]) and r[4] > 0 and r[5] > 0, int(r[1]), int(r[2]), int(r[3])) for r in rows]
        if self.method == "orb":
            rows = self.eng.attempt_orb_batch(jobs, self.params, getattr(self, "orbMaxDistance", -1), self.offsetEvaluate)
            return [(bool(r[0]) and r[4] > 0 and r[5] > 0, int(r[1]), int(r[2]), int(r[3])) for r in rows]
        if self.method == "phase":
            rows = self.eng.attempt_phase_batch(jobs)
            # offset = [int(y), int(x)] (truncation); accepted when response > threshold (Stitcher.py:231-236)
            return [(bool(r[2] > self.phaseThr), int(r[1]), int(r[0]), 0) for r in rows]
        raise ValueError("method %r" % (self.method,))

    def _surf_batch(self, jobs):
        """Fused batch with an adaptive keypoint capacity: kernels are launched over capacity-sized grids (no host
        sync inside a batch), so the capacity follows the largest ROI seen so far (x1.5 + 1024); an overflow falls
        back to the library default (h*w/24 + 4096) and repeats the batch."""
        cap = getattr(self, "_kp_cap", 0)
        adaptive = hasattr(self.eng, "set_keypoint_capacity")
        try:
            # the override is in force only around this registrar's own batches: other users of the engine keep the default
            if adaptive and cap:
                self.eng.set_keypoint_capacity(cap)
            try:
                rows = self._surf_call(jobs)
            except Exception:
                if not cap:
                    raise
                self.capacity_retries = getattr(self, "capacity_retries", 0) + 1      # reported by bench.py
                self._kp_cap = 0
                self._kp_seen = 0
                self.eng.set_keypoint_capacity(0)
                rows = self._surf_call(jobs)
        finally:
            if adaptive and cap:
                self.eng.set_keypoint_capacity(0)
        if adaptive and len(rows):
            seen = max(int(rows[:, 4:6].max()), getattr(self, "_kp_seen", 0))
            self._kp_seen = seen
            want = int(seen * 1.5) + 1024
            cap = getattr(self, "_kp_cap", 0)
            if seen > 0 and want != cap and (cap == 0 or want > cap or want < cap * 0.6):
                self._kp_cap = want
        return rows

    def _surf_call(self, jobs):
        if self.enhance[0]:
            return self.eng.attempt_surf_batch_enhanced(jobs, self.params, self.searchRatio, self.offsetEvaluate, self.enhance)
        return self.eng.attempt_surf_batch(jobs, self.params, self.searchRatio, self.offsetEvaluate)

    def _correct(self, raw, d, i, shapeA, shapeB):
        """Stitcher.py:352-360: ROI-relative vote -> full-tile offset."""
        dx, dy = raw
        if d == 1:
            dx = dx + shapeA[0] - int(i * self.roiRatio * shapeA[0])
        elif d == 2:
            dy = dy + shapeA[1] - int(i * self.roiRatio * shapeA[1])
        elif d == 3:
            dx = dx - (shapeB[0] - int(i * self.roiRatio * shapeB[0]))
        elif d == 4:
            dy = dy - (shapeB[1] - int(i * self.roiRatio * shapeB[1]))
        return dx, dy

    # -- sequentially-equivalent chain over pairs [first, last) ------------------------------------------------------
    def chain(self, handles, shapes, first, last, d_in, memo=None, cache=None, midpath=False, stop_on_fail=False, hint=None):
        """-> (int32[last-first, 6], d_out).

        An attempt is a pure function of (pair, direction, i), so WHICH attempts are evaluated together is free;
        the result is always selected in the reference's candidate order.  What is batched is chosen by a small
        predictor fed with the history of this chain: the length of the run of pairs that kept the direction
        (shooting paths are serpentines: long run, turn, long run, ...) bounds the speculation window, a predicted
        turn gets its whole first candidate ring in one batch, and the ring position that resolved the last turn
        from the same incoming direction bounds the first resolve batch.
        memo: {(k, d): (row, d_next)} and cache: {(k, d, i): attempt} may be shared between chains."""
        memo = {} if memo is None else memo
        cache = {} if cache is None else cache
        out = np.zeros((last - first, RESULT_INTS), np.int32)

        def evaluate(items):
            todo = [it for it in dict.fromkeys(items) if it not in cache and it[0] < last]
            if todo:
                for it, r in zip(todo, self._attempts(handles, shapes, todo)):
                    cache[it] = r

        runs, run_len, slow, ring_hint = [], 0, 1, {}
        trans2 = {}                                   # (direction before, direction) -> direction the next turn led to
        prev_d = 0
        if hint is not None and len(hint) and first > 0:
            # a chain that starts inside the path: prime the predictor with the history the predicted directions imply for the pairs
            # before `first` (csrc/grid.hip does the same); bookkeeping only
            hd = int(hint[0])
            for kk in range(min(first, len(hint))):
                nd = int(hint[kk])
                if not (1 <= hd <= 4 and 1 <= nd <= 4):
                    break
                if nd == hd:
                    run_len += 1
                    slow = min(2 * slow, self.window)
                else:
                    runs.append(run_len)
                    run_len, slow = 1, 1
                    trans2[(prev_d, hd)] = nd
                    ring0 = [c[0] for c in self.rings(hd)[0]]
                    if nd in ring0:
                        ring_hint[hd] = ring0.index(nd)
                    prev_d = hd
                hd = nd
            if hd != d_in:                            # entered differently than predicted: no basis
                runs, run_len, slow, ring_hint, trans2, prev_d = [], 0, 1, {}, {}, 0
        d = d_in
        k = first

        def plan(k0, d0, p0):
            """Predicted continuation of the path as one batch (up to `window` attempts): the rest of the current run, the
            candidate ring of the predicted turn up to the direction it led to last time, the following run(s), ..."""
            items, R, rl, cd, cp, kk = [], list(runs), run_len, d0, p0, k0
            while kk < last and len(items) < self.window:
                pred = R[-2] if len(R) >= 2 else None
                if pred is None:
                    break
                remaining = pred - rl
                if remaining < 0:
                    break                                  # this run already outlived the prediction: no basis for a turn, slow start instead
                if remaining >= 1:
                    n = min(remaining, self.window - len(items), last - kk)
                    items += [(kk + t, cd, 1) for t in range(n)]
                    kk += n; rl += n
                    if n < remaining:
                        break
                    continue
                ring = self.rings(cd)[0]
                nd = trans2.get((cp, cd))
                if nd is None or all(c[0] != nd for c in ring):
                    items += [(kk,) + c for c in ring[:ring_hint.get(cd, len(ring) - 1) + 1]]
                    break
                upto = [c[0] for c in ring].index(nd)
                items += [(kk,) + c for c in ring[:upto + 1]]
                R.append(rl); rl = 1
                cp, cd = cd, nd
                kk += 1
            return items

        def plan_hint(k0, d0):
            """The predicted directions as the plan itself: the run at the current direction up to the predicted change, the candidate ring
            of that pair up to the predicted new direction, the next run, ... -- what the history-driven plan arrives at after two
            serpentine periods, available from the first pair on."""
            items, cd, kk = [], d0, k0
            while kk < last and kk < len(hint) and len(items) < self.window:
                hd = int(hint[kk])
                if not 1 <= hd <= 4:
                    break
                ring = self.rings(cd)[0]
                ds = [c[0] for c in ring]
                if hd == cd or hd not in ds:
                    items.append((kk, cd, 1))
                else:
                    items += [(kk,) + c for c in ring[:ds.index(hd) + 1]]
                    cd = hd
                kk += 1
            return items

        while k < last:
            if (k, d) in memo:
                row, d_next = memo[(k, d)]
            else:
                rings = self.rings(d)
                if (k, d, 1) not in cache:
                    items = plan_hint(k, d) if hint is not None and len(hint) else []
                    if not items:
                        items = plan(k, d, prev_d)
                    if not items:                              # no history yet: slow start
                        items = [(kk, d, 1) for kk in range(k, min(k + slow, last)) if (kk, d) not in memo]
                    evaluate(items)
                    if (k, d, 1) not in cache:
                        evaluate([(k, d, 1)])
                found = None
                for ri, ring in enumerate(rings):
                    pos = 0
                    while pos < len(ring) and found is None:
                        if (k,) + ring[pos] not in cache:
                            h = ring_hint.get(d, len(ring) - 1) if ri == 0 else len(ring) - 1
                            stop = max(pos, min(h, len(ring) - 1))
                            evaluate([(k,) + c for c in ring[pos:stop + 1]])
                        st, a, b, v = cache[(k,) + ring[pos]]
                        if st:
                            found = ring[pos] + (a, b, v)
                            if ri == 0:
                                ring_hint[d] = pos
                        pos += 1
                    if found is not None:
                        break
                if found is not None:
                    dd, ii, a, b, v = found
                    dx, dy = self._correct((a, b), dd, ii, shapes[k], shapes[k + 1])
                    row = np.array([1, dx, dy, dd, ii, v], np.int32)
                    d_next = dd                       # self.direction = localDirection
                else:
                    row = np.array([0, 0, 0, d, 0, 0], np.int32)
                    d_next = d                        # a failed pair leaves self.direction untouched
                memo[(k, d)] = (row, d_next)
            # predictor bookkeeping
            if row[0] and d_next == d:
                run_len += 1
                # a chain that starts in the middle of a path sees a truncated run and a turn soon after: until it has seen two
                # runs, an overshoot past that turn is all waste, so it speculates at most 4 pairs ahead
                slow = min(2 * slow, 4 if (midpath and len(runs) < 2) else self.window)
            elif row[0]:
                runs.append(run_len)
                run_len, slow = 1, 1
                trans2[(prev_d, d)] = d_next
                prev_d = d
            out[k - first] = row
            d = d_next
            k += 1
            if stop_on_fail and not row[0]:
                break                             # flowStitch discards everything behind the first break (Stitcher.py:74-76)
        return out, d

    native = True      # run whole chains inside the library (vfsms_pairs_offsets) when the engine offers it; chain() is the same machine in Python

    def _grid_params(self, hint=None):
        kw = dict(hint=hint) if hint is not None else {}
        return self.eng.grid_params(method=self.method, roiRatio=self.roiRatio, searchRatio=self.searchRatio, offsetEvaluate=self.offsetEvaluate,
                                    directIncre=self.directIncre, window=self.window, surf=self.params if self.method == "surf" else None,
                                    orb=self.params if self.method == "orb" else None, phaseResponseThreshold=self.phaseThr,
                                    orbMaxDistance=getattr(self, "orbMaxDistance", -1), enhance=self.enhance, **kw)

    def _native_stats(self, st):
        self.stats["attempts"] += st[0]; self.stats["batches"] += st[1]
        self.capacity_retries = getattr(self, "capacity_retries", 0) + st[2]
        self.stats["sum_nq_nt"] += st[3]; self.stats["sum_nq_plus_nt"] += st[4]; self.stats["roi_px"] += st[5]

    def _prediction(self, P, hint):
        """the caller's hint, else what the last path of the same length taught this registrar"""
        if hint is not None:
            return hint
        m = self.path_memory
        return m if (self.remember and m is not None and len(m) == P) else None

    MISPREDICT_EXTRA = 0.25     # a prediction is "wrong" when it cost more than this fraction of the attempts it had promised

    def _learn(self, table, predicted=None, direction_in=1):
        """The path just registered becomes the prediction of the next one of its length -- unless the prediction it was registered WITH
        (`predicted`: the memory, never a caller's hint) turned out wrong.  Wrong = the extra attempts it caused, estimated from the table
        every rank holds (per pair whose accepted direction differs from the predicted one: the rotation steps between the two, at least
        one), exceed MISPREDICT_EXTRA of the attempts it predicted (path_weights).  One misprediction: the new pattern is adopted on
        probation (a session that moves on to another scan pattern is primed again from the second path on).  Two in a row (paths that do not
        repeat): the memory is dropped, the next path runs cold -- a stale prior costs attempts (75 against 43 on the
        dendriticCrystal neighbourhoods), a cold path only costs batches.  Decided from the table alone, so every rank of the sharded form
        takes the same decision."""
        if not (self.remember and len(table)):
            return
        new = [int(r[3]) if 1 <= int(r[3]) <= 4 else 1 for r in table]
        if predicted is not None and len(predicted) == len(new):
            extra = 0
            for r, h, a in zip(table, predicted, new):
                if int(r[0]) == 1 and int(h) != a:
                    steps, c = 0, int(h)
                    while c != a and steps < 4 and self.directIncre != 0:
                        c = _rotate(c, self.directIncre); steps += 1
                    extra += max(steps, 1)
            promised = sum(self.path_weights(predicted, int(direction_in)))
            if extra > self.MISPREDICT_EXTRA * promised:
                self.mispredictions += 1
                if self.path_suspect:
                    self.path_memory, self.path_suspect = None, False
                    return
                self.path_suspect = True
            else:
                self.path_suspect = False
        else:
            self.path_suspect = False             # no memory was on trial (first path, another length, a caller's hint): what is learned now starts clean
        self.path_memory = new

    def _memory_prediction(self, P, hint):
        """the memory, when it is what _prediction(P, hint) hands out (None for a caller's hint: only the memory is put on trial)"""
        m = self.path_memory
        return m if (hint is None and self.remember and m is not None and len(m) == P) else None

    def register(self, handles, shapes, direction=1, stop_on_fail=False, hint=None):
        """All P = len(handles)-1 consecutive pairs on this GPU.  -> (int32[P, 6], final direction).
        stop_on_fail: stop behind the first pair that cannot be registered (rows after it stay zero).
        hint: predicted accepted directions (default: the path memory)."""
        P = len(handles) - 1
        mem = self._memory_prediction(P, hint)
        hint = self._prediction(P, hint)
        if self.native and hasattr(self.eng, "pairs_offsets"):
            out, d, st = self.eng.pairs_offsets(handles, shapes, self._grid_params(hint), 0, P, direction, False, stop_on_fail)
            self._native_stats(st)
        else:
            out, d = self.chain(handles, shapes, 0, P, direction, stop_on_fail=stop_on_fail, hint=hint)
        if not stop_on_fail or bool(np.all(out[:, 0] == 1)):
            self._learn(out, mem, direction)
        return out, d

    # -- pair-sharded ---------------------------------------------------------------------------------------------------
    BLIND_START_COST = 3.0     # a chunk entered with an unknown direction tries four first candidates instead of one

    @staticmethod
    def chunk_bounds(n_pairs, world, weights=None, blind_cost=None):
        """Contiguous chunks of the path, one per rank.  Without `weights`: balanced by pair count (the remainder goes to the lowest ranks;
        rank 0 is the cheapest, it knows its incoming direction).  With weights (expected attempts per pair, path_weights): the partition
        that minimises the most loaded rank (every rank but 0 carrying `blind_cost`, the price of a start with an unknown direction) --
        a pair that changes the direction costs 2-4 attempts, so chunks by pair count leave the ranks that hold the turns over the mean.
        Deterministic: every rank computes the same."""
        if weights is None or world <= 1 or n_pairs <= 0:
            base, extra = divmod(n_pairs, world)
            bounds, lo = [], 0
            for r in range(world):
                hi = lo + base + (1 if r < extra else 0)
                bounds.append((lo, hi))
                lo = hi
            return bounds
        w = [float(v) for v in weights]
        if len(w) != n_pairs:
            raise ValueError("chunk_bounds: one weight per pair")
        blind = GridRegistrar.BLIND_START_COST if blind_cost is None else float(blind_cost)

        def cuts(limit):
            """greedy: every rank takes pairs while it stays within `limit` -> bounds, or None when the path does not fit"""
            bounds, lo = [], 0
            for r in range(world):
                acc, hi = (blind if r > 0 else 0.0), lo
                while hi < n_pairs and acc + w[hi] <= limit:
                    acc += w[hi]; hi += 1
                bounds.append((lo, hi))
                lo = hi
            return bounds if lo == n_pairs else None
        lo_l, hi_l = 0.0, sum(w) + blind
        for _ in range(60):                                 # bisection on the bottleneck load
            mid = 0.5 * (lo_l + hi_l)
            if cuts(mid) is None:
                lo_l = mid
            else:
                hi_l = mid
        return cuts(hi_l)

    def path_weights(self, directions, direction_in=1):
        """Expected attempts per pair from a PREDICTION of the accepted directions (e.g. the stage's scan pattern: a column serpentine of
        known height): a pair that keeps the direction costs one attempt, a pair that changes it the candidates the rotation tries up to
        the new direction (Stitcher.py:319-351: 1 + rotation steps) -- what a chain with a primed predictor evaluates (BASELINE
        configs[1]'s serpentine: 89 pairs, 16 direction changes, 123 attempts).  Only the work split depends on it."""
        w, d = [], int(direction_in)
        for nd in directions:
            nd = int(nd)
            steps, c = 0, d
            while c != nd and steps < 4 and self.directIncre != 0:
                c = _rotate(c, self.directIncre); steps += 1
            w.append(1.0 + steps)
            d = nd
        return w

    def _bounds(self, P, world, weights, hint, direction):
        """the work split of the sharded form: by the caller's weights, else -- with a hint -- by the attempts the hint predicts (a hinted
        start costs nothing extra), else by pair count"""
        # (the same split is asked for twice per registered path -- payload and assembly -- and path after path: the bisection is 60 greedy
        #  passes over the pairs, half a millisecond of interpreter time that a rank's 6-ms step does not have)
        key = (P, world, None if weights is None else tuple(weights), None if hint is None else tuple(int(v) for v in hint), int(direction), self.directIncre)
        cached = self.__dict__.get("_bounds_cache")
        if cached is not None and cached[0] == key:
            return cached[1]
        if weights is None and hint is not None and len(hint) == P and world > 1:
            out = self.chunk_bounds(P, world, self.path_weights(hint, direction), blind_cost=0.0)
        else:
            out = self.chunk_bounds(P, world, weights)
        self._bounds_cache = (key, out)
        return out

    def shard_payload(self, handles, shapes, direction, rank, world, weights=None, hint=None, blind=False):
        """This rank's offset table: int32[4 * per * 6 + 4] = results for each possible incoming direction
        (only the true one on rank 0 / when directIncre == 0) followed by the direction each chain ends in (0: chain not evaluated).
        hint: predicted accepted direction of every pair (the stage's scan pattern).  A rank > 0 then follows ONLY the chain entered with
        the direction the hint gives its predecessor pair -- one first candidate instead of four; a wrong hint is found out by assemble()
        (the chain it needs is marked missing) and repaired by register_sharded in a second round.  Results never depend on the hint."""
        P = len(shapes) - 1
        bounds = self._bounds(P, world, weights, hint, direction)
        lo, hi = bounds[rank]
        per = max(b - a for a, b in bounds)
        dirs = [direction] if (self.directIncre == 0 or rank == 0) else [1, 2, 3, 4]
        if len(dirs) > 1 and hint is not None and not blind and 0 < lo <= len(hint) and int(hint[lo - 1]) in (1, 2, 3, 4):
            dirs = [int(hint[lo - 1])]
        table = np.zeros((4, per, RESULT_INTS), np.int32)
        d_out = np.zeros(4, np.int32)
        memo, cache = {}, {}
        if len(dirs) > 1 and hi > lo and self.native and hasattr(self.eng, "pairs_offsets_blind"):
            # the four blind chains inside the library (csrc/grid.hip: the same machine, shared cache and memo)
            res, dn, st = self.eng.pairs_offsets_blind(handles, shapes, self._grid_params(), lo, hi, per)
            self._native_stats(st)
            return np.concatenate([np.asarray(res, np.int32).reshape(-1), np.asarray(dn, np.int32)])
        if len(dirs) > 1 and hi > lo:
            # the incoming direction is unknown here: every chain needs its own first candidate of the first pair, so all four
            # are evaluated as one batch instead of being discovered one chain after the other
            for it, r in zip([(lo, d, 1) for d in dirs], self._attempts(handles, shapes, [(lo, d, 1) for d in dirs])):
                cache[it] = r
        for d_in in dirs:
            if hi > lo and len(dirs) == 1 and self.native and hasattr(self.eng, "pairs_offsets"):
                res, dn, st = self.eng.pairs_offsets(handles, shapes, self._grid_params(hint), lo, hi, d_in, rank > 0, False)
                self._native_stats(st)
                table[d_in - 1, :hi - lo] = res
            elif hi > lo:
                res, dn = self.chain(handles, shapes, lo, hi, d_in, memo, cache, midpath=rank > 0, hint=hint if len(dirs) == 1 else None)
                table[d_in - 1, :hi - lo] = res
            else:
                dn = d_in
            d_out[d_in - 1] = dn
        return np.concatenate([table.reshape(-1), d_out])

    def assemble(self, gathered, n_pairs, world, direction, weights=None, missing=None, hint=None):
        """Walk the gathered tables rank by rank, selecting the chain consistent with the true incoming direction.
        missing (a list): receives the ranks whose table lacks the chain the walk needs (a hinted start that guessed wrong); the walk
        stops at the first of them and the result is then incomplete."""
        P = n_pairs
        bounds = self._bounds(P, world, weights, hint, direction)
        per = max(b - a for a, b in bounds)
        full = np.zeros((P, RESULT_INTS), np.int32)
        d = direction
        for r in range(world):
            a, b = bounds[r]
            t = np.asarray(gathered[r][:-4]).reshape(4, per, RESULT_INTS)
            dn = gathered[r][-4:]
            if b > a and int(dn[d - 1]) == 0:
                if missing is None:
                    raise RuntimeError("rank %d did not evaluate the chain entered with direction %d" % (r, d))
                missing.append(r)
                return full, d
            full[a:b] = t[d - 1, :b - a]
            d = int(dn[d - 1]) if b > a else d
        return full, d

    def register_sharded(self, handles, shapes, direction, rank, world, all_gather, weights=None, hint=None):
        """handles/shapes are indexed by GLOBAL tile index (only this rank's chunk + halo need be valid).
        all_gather(int32 ndarray [C]) -> int32 ndarray [world, C]   (the single collective of the path).
        weights: expected attempts per pair (path_weights) for the work split; hint: predicted accepted directions (shard_payload);
        every rank must pass the same.  With a hint that turns out wrong for some rank, ONE repair round follows: the ranks whose
        chunk was entered with another direction than they assumed register it again for all four (the blind form) and the tables are
        gathered a second time -- every rank sees the same tables, so all of them take the same decision.
        Returns the same (int32[P, 6], final direction) on every rank."""
        P = len(shapes) - 1
        mem = self._memory_prediction(P, hint)
        hint = self._prediction(P, hint)
        payload = self.shard_payload(handles, shapes, direction, rank, world, weights, hint)
        gathered = all_gather(payload)
        if len(gathered) != world:
            raise RuntimeError("all_gather returned %d payloads for a world of %d ranks" % (len(gathered), world))
        if hint is None:
            full, d = self.assemble(gathered, P, world, direction, weights)
            self._learn(full, mem, direction)
            return full, d
        missing = []
        full, d = self.assemble(gathered, P, world, direction, weights, missing, hint)
        if not missing:
            self._learn(full, mem, direction)
            return full, d
        # repair: every rank that followed a single hinted chain is a suspect (a wrong direction upstream changes what enters the ranks
        # behind it); those whose assumption is not confirmed by the first walk redo their chunk blind.  One extra collective.
        self.hint_repairs = getattr(self, "hint_repairs", 0) + 1
        gathered = np.array(gathered, np.int32, copy=True)
        confirmed = set(range(missing[0]))                   # the walk reached these ranks with the direction they had assumed
        if rank in confirmed or rank == 0:
            mine = payload
        else:                                                # the same chunk (the hint's work split), now for every incoming direction
            mine = self.shard_payload(handles, shapes, direction, rank, world, weights, hint, blind=True)
        gathered = all_gather(mine)
        full, d = self.assemble(gathered, P, world, direction, weights, None, hint)
        self._learn(full, mem, direction)
        return full, d

    def register_projected(self, handles, shapes, direction, world, probe=None, all_gather=None):
        """The sharded form of `world` ranks on ONE device, one rank AFTER the other, through the code the ranks run (shard_payload with
        the rank's chunk, hinted start and plan; assemble; the repair round; _learn) -- what a development box with one GPU can measure of
        an N-GPU step: every emulated rank has the device to itself, as it would on its own GPU.
        probe(): called after every rank's part -> anything (bench.py: the HIP-event milliseconds since the last call); all_gather: run on
        the stacked payloads once per round (bench.py: the RCCL gather of a world of one, for its latency), default none.
        -> (table, final direction, per_rank [dict(pairs, wall_s, attempts, batches, probe, repair_wall_s)], tail_s): a rank's step is its
        wall_s (+ repair_wall_s) + tail_s (gather + assemble + learn, common to all ranks)."""
        import time
        P = len(shapes) - 1
        mem = self._memory_prediction(P, hint=None)
        hint = self._prediction(P, None)
        bounds = self._bounds(P, world, None, hint, direction)
        per_rank, payloads = [], []
        for r in range(world):
            a0, b0 = self.stats["attempts"], self.stats["batches"]
            t0 = time.perf_counter()
            payloads.append(self.shard_payload(handles, shapes, direction, r, world, None, hint))
            wall = time.perf_counter() - t0
            per_rank.append(dict(rank=r, pairs=bounds[r][1] - bounds[r][0], wall_s=wall, attempts=self.stats["attempts"] - a0,
                                 batches=self.stats["batches"] - b0, probe=probe() if probe else None, repair_wall_s=0.0))
        t0 = time.perf_counter()
        gathered = np.stack(payloads)
        if all_gather is not None:
            all_gather(payloads[0])
        missing = []
        if hint is None:
            full, d = self.assemble(gathered, P, world, direction, None)
        else:
            full, d = self.assemble(gathered, P, world, direction, None, missing, hint)
        tail = time.perf_counter() - t0
        if missing:
            self.hint_repairs = getattr(self, "hint_repairs", 0) + 1
            confirmed = set(range(missing[0]))
            for r in range(1, world):
                if r in confirmed:
                    continue
                a0, b0 = self.stats["attempts"], self.stats["batches"]
                t0 = time.perf_counter()
                payloads[r] = self.shard_payload(handles, shapes, direction, r, world, None, hint, blind=True)
                per_rank[r]["repair_wall_s"] = time.perf_counter() - t0
                per_rank[r]["attempts"] += self.stats["attempts"] - a0; per_rank[r]["batches"] += self.stats["batches"] - b0
                if probe:
                    per_rank[r]["probe_repair"] = probe()
            t0 = time.perf_counter()
            if all_gather is not None:
                all_gather(payloads[0])
            full, d = self.assemble(np.stack(payloads), P, world, direction, None, None, hint)
            tail += time.perf_counter() - t0
        t0 = time.perf_counter()
        self._learn(full, mem, direction)
        tail += time.perf_counter() - t0
        return full, d, per_rank, tail


def serpentine_directions(rows, cols, first=1, across=2):
    """Predicted accepted directions of a column-major serpentine of rows x cols tiles (rows - 1 pairs down, one across, rows - 1 up, ...):
    a scan-pattern hint for GridRegistrar.register(hint=...) / Stitcher.pathHint.  first: direction of the first column (1: the next tile
    lies below), across: direction of the step to the next column."""
    back = {1: 3, 3: 1, 2: 4, 4: 2}[first]
    out = []
    for c in range(cols):
        out += [first if c % 2 == 0 else back] * (rows - 1)
        if c < cols - 1:
            out.append(across)
    return out


def split_segments(results):
    """flowStitchWithMutiple's segmentation (Stitcher.py:96-127) from a per-pair result table:
    -> [(first tile, last tile inclusive, [[dx, dy], ...])]."""
    segs = []
    start = 0
    offs = []
    P = len(results)
    for k in range(P):
        if results[k][0]:
            offs.append([int(results[k][1]), int(results[k][2])])
        else:
            segs.append((start, k, offs))
            start, offs = k + 1, []
    segs.append((start, P, offs))
    return segs

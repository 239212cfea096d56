"""CPU tests: the speculative batched registrar and its pair-sharded form give EXACTLY the sequential result of
Stitcher.calculateOffsetForFeatureSearchIncre applied pair after pair (direction state threaded through),
for random truth tables with failures, late successes (i > 1) and false-positive directions."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import imagestitch_amd as isa
from imagestitch_amd.grid import GridRegistrar, split_segments
from scripted import ScriptedAttemptEngine, random_truth

SHAPE = (1000, 1400)


class SeqStitcher(isa.Stitcher):
    def __init__(self, eng):
        self.eng2 = eng

    def _featureAttempt(self, imageA, imageB, direction, searchRatio):
        ra = isa.roi_rect(SHAPE, direction, "first", searchRatio); rb = isa.roi_rect(SHAPE, direction, "second", searchRatio)
        row = self.eng2.attempt_surf_batch([(self.k, self.k + 1, ra[0], ra[1], rb[0], rb[1], ra[2], ra[3])])[0]
        return (bool(row[0]), [int(row[1]), int(row[2])])


def sequential(accept, roiRatio, incre, d0):
    eng = ScriptedAttemptEngine(SHAPE, roiRatio, accept)
    s = SeqStitcher(eng); s.isPrintLog = False
    s.roiRatio, s.directIncre, s.direction = roiRatio, incre, d0
    A = np.zeros(SHAPE, np.uint8)
    rows = []
    for k in range(len(accept)):
        s.k = k
        st, off = s.calculateOffsetForFeatureSearchIncre([A, A])
        rows.append([1, off[0], off[1], s.direction] if st else [0, 0, 0, s.direction])
    return rows, s.direction, len(eng.log)


@pytest.mark.parametrize("seed", range(12))
def test_chain_and_shards_equal_sequential(seed):
    rng = np.random.default_rng(seed)
    roiRatio = float(rng.choice([0.1, 0.2]))
    incre = int(rng.choice([-1, 0, 1]))
    d0 = int(rng.integers(1, 5))
    P = int(rng.integers(1, 40))
    accept = random_truth(rng, P, roiRatio)
    seq, d_end, n_seq = sequential(accept, roiRatio, incre, d0)
    handles = list(range(P + 1)); shapes = [SHAPE] * (P + 1)
    for window in (1, 4, 16):
        eng = ScriptedAttemptEngine(SHAPE, roiRatio, accept)
        reg = GridRegistrar(eng, roiRatio=roiRatio, directIncre=incre, window=window)
        res, d = reg.register(handles, shapes, d0)
        assert d == d_end
        assert [list(r[:4]) for r in res.tolist()] == seq, (seed, window)
        for world in (1, 2, 3, 5):
            payloads = []
            for rank in range(world):
                e2 = ScriptedAttemptEngine(SHAPE, roiRatio, accept)
                r2 = GridRegistrar(e2, roiRatio=roiRatio, directIncre=incre, window=window)
                payloads.append(r2.shard_payload(handles, shapes, d0, rank, world))
            full, d2 = reg.assemble(np.stack(payloads), P, world, d0)
            assert d2 == d_end and np.array_equal(full, res), (seed, window, world)


def test_speculation_never_reorders_candidates():
    # pair 1 truly lies in direction 2 but ALSO passes in direction 1 at i=2: the sequential search (ini 1, incre 1)
    # tries (1,1) (2,1) ... and must accept (2,1) first; a registrar that batched by ring would still agree
    accept = [{(1, 1): (3, 3)}, {(2, 1): (4, 4), (1, 2): (9, 9)}, {(2, 1): (5, 5)}]
    seq, d_end, _ = sequential(accept, 0.2, 1, 1)
    eng = ScriptedAttemptEngine(SHAPE, 0.2, accept)
    res, d = GridRegistrar(eng, roiRatio=0.2, directIncre=1, window=8).register(list(range(4)), [SHAPE] * 4, 1)
    assert [list(r[:4]) for r in res.tolist()] == seq and d == d_end == 2
    assert res[1][4] == 1 and res[1][3] == 2


def test_split_segments_matches_flow_restart_arithmetic():
    rows = [[1, 5, 0], [0, 0, 0], [1, 6, 0], [1, 7, 0], [0, 0, 0]]
    assert split_segments(rows) == [(0, 1, [[5, 0]]), (2, 4, [[6, 0], [7, 0]]), (5, 5, [])]


def test_two_process_gloo_all_gather(tmp_path):
    """world_size 2 over torch.distributed / gloo on CPU: the N > 1 code path of bench.py's collective."""
    out = os.path.join(str(tmp_path), "res.json")
    worker = os.path.join(os.path.dirname(__file__), "dist_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", worker, out]
    subprocess.check_call(cmd, env=env, timeout=600)
    got = json.load(open(out))
    rng = np.random.default_rng(77)
    accept = random_truth(rng, 23, 0.2)
    seq, d_end, _ = sequential(accept, 0.2, 1, 1)
    assert got["direction"] == d_end and [r[:4] for r in got["rows"]] == seq


def test_eight_process_gloo_config4_path(tmp_path):
    """BASELINE configs[4]'s registration half on scripted attempts: the 1023 pairs of a 32 x 32 serpentine sharded over world_size 8
    (gloo, CPU): contiguous chunks, blind chunk starts, ONE all-gather; every rank assembles the sequential result.  Also records how
    the work spread (attempts / batches per rank) -- the quantity that bounds strong scaling."""
    from scripted import serpentine_truth
    out = os.path.join(str(tmp_path), "res8.json")
    worker = os.path.join(os.path.dirname(__file__), "dist_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29617", worker, out, "1023", "5", "24"]
    subprocess.check_call(cmd, env=env, timeout=900)
    got = json.load(open(out))
    accept = serpentine_truth(32, 32, 0.2)
    assert len(accept) == 1023
    seq, d_end, _ = sequential(accept, 0.2, 1, 1)
    assert got["direction"] == d_end and [r[:4] for r in got["rows"]] == seq
    print("attempts per rank", got["attempts"], "batches per rank", got["batches"])
    assert len(got["attempts"]) == 8 and sum(got["attempts"]) < 1.4 * 1100 and max(got["attempts"]) < 1.35 * sum(got["attempts"]) / 8


def test_native_pairs_offsets_state_machine_equals_python_and_sequential():
    """vfsms_pairs_offsets_eval -- the candidate state machine inside libvfsms.so (csrc/grid.hip), fed by a scripted evaluator through
    a C callback (no GPU): on random truth tables with failures, late successes and false-positive directions it must return the
    rows of the sequential search AND evaluate exactly the batches the Python registrar evaluates (same attempts, same order)."""
    import imagestitch_amd as isa
    from imagestitch_amd._lib import pairs_offsets_eval, Engine
    for seed in range(40):
        rng = np.random.default_rng(1000 + seed)
        roiRatio = float(rng.choice([0.1, 0.2]))
        incre = int(rng.choice([-1, 0, 1]))
        P = int(rng.integers(1, 40))
        window = int(rng.choice([1, 3, 8, 24]))
        d0 = int(rng.integers(1, 5))
        accept = random_truth(rng, P, roiRatio)
        seq, d_end, _ = sequential(accept, roiRatio, incre, d0)
        eng = ScriptedAttemptEngine(SHAPE, roiRatio, accept)
        reg = GridRegistrar(eng, roiRatio=roiRatio, directIncre=incre, window=window)
        res_py, d_py = reg.chain(list(range(P + 1)), [SHAPE] * (P + 1), 0, P, d0)
        log = []

        def attempts(items, accept=accept, log=log):
            rows = []
            for (k, d, i) in items:
                log.append((k, d, i))
                acc = accept[k]
                ok = (d, i) in acc
                raw = acc[(d, i)] if ok else (7, -3)
                rows.append([int(ok), raw[0], raw[1], 5 if ok else 1, 100, 100, 10, 0])
            return rows
        params = Engine.grid_params(method="surf", roiRatio=roiRatio, directIncre=incre, window=window)
        res_c, d_c, st = pairs_offsets_eval(attempts, [SHAPE] * (P + 1), params, 0, P, d0)
        assert [list(r[:4]) for r in res_c.tolist()] == seq and d_c == d_end, (seed, res_c.tolist(), seq)
        assert np.array_equal(res_c, res_py) and d_c == d_py
        assert log == eng.log and st[0] == len(log) == reg.stats["attempts"] and st[1] == reg.stats["batches"], (seed, len(log), len(eng.log))
    # stop_on_fail: nothing behind the first break is reported
    accept = [{(1, 1): (3, 3)}, {}, {(1, 1): (4, 4)}]
    res, d, _st = pairs_offsets_eval(lambda items: [[int((dd, i) in accept[k])] + ([3, 3, 5] if (dd, i) in accept[k] else [0, 0, 0]) + [9, 9, 4, 0] for (k, dd, i) in items],
                                     [SHAPE] * 4, Engine.grid_params(roiRatio=0.2, directIncre=1, window=8), 0, 3, 1, False, True)
    assert res[:, 0].tolist() == [1, 0, 0]


def test_native_blind_chains_equal_python_twin():
    """vfsms_pairs_offsets_blind_eval (csrc/grid.hip) -- the chunk of a rank > 0, registered for all four possible incoming directions with a
    shared attempt cache and (pair, direction) memo -- against GridRegistrar.shard_payload's Python twin on random truth tables: the same
    payload (4 x per x 6 rows + 4 end directions), the same attempts in the same order, the same batch count; and the chain of the TRUE
    incoming direction equals the sequential search of that chunk."""
    from imagestitch_amd._lib import pairs_offsets_blind_eval, Engine
    for seed in range(30):
        rng = np.random.default_rng(7000 + seed)
        roiRatio = float(rng.choice([0.1, 0.2]))
        incre = int(rng.choice([-1, 1]))
        P = int(rng.integers(4, 60))
        world = int(rng.integers(2, 5))
        rank = int(rng.integers(1, world))
        window = int(rng.choice([3, 8, 24]))
        accept = random_truth(rng, P, roiRatio)
        eng = ScriptedAttemptEngine(SHAPE, roiRatio, accept)
        reg = GridRegistrar(eng, roiRatio=roiRatio, directIncre=incre, window=window)
        reg.native = False
        shapes = [SHAPE] * (P + 1)
        pay_py = reg.shard_payload(list(range(P + 1)), shapes, 1, rank, world)
        bounds = GridRegistrar.chunk_bounds(P, world)
        lo, hi = bounds[rank]
        per = max(b - a for a, b in bounds)
        log = []

        def attempts(items, accept=accept, log=log):
            rows = []
            for (k, d, i) in items:
                log.append((k, d, i))
                acc = accept[k]
                ok = (d, i) in acc
                raw = acc[(d, i)] if ok else (7, -3)
                rows.append([int(ok), raw[0], raw[1], 5 if ok else 1, 100, 100, 10, 0])
            return rows
        params = Engine.grid_params(method="surf", roiRatio=roiRatio, directIncre=incre, window=window)
        res, dn, st = pairs_offsets_blind_eval(attempts, shapes, params, lo, hi, per)
        pay_c = np.concatenate([np.asarray(res, np.int32).reshape(-1), np.asarray(dn, np.int32)])
        assert np.array_equal(pay_c, pay_py), (seed, lo, hi)
        assert log == eng.log and st[0] == len(log) == reg.stats["attempts"] and st[1] == reg.stats["batches"], (seed, len(log), len(eng.log))
        for d_in in (1, 2, 3, 4):                              # every chain is the sequential search of the chunk entered with d_in
            seq, d_end, _ = sequential(accept[lo:hi], roiRatio, incre, d_in)
            assert [list(r[:4]) for r in res[d_in - 1][:hi - lo].tolist()] == seq and int(dn[d_in - 1]) == d_end, (seed, d_in)


def _lockstep_sharded(accept, roiRatio, incre, d0, world, window, hint, native=False):
    """register_sharded on `world` threads with an in-process all-gather (a barrier + a shared list): the collective protocol -- including
    the repair round of a wrong hint -- exactly as the ranks of a torch.distributed job run it"""
    import threading
    P = len(accept)
    handles, shapes = list(range(P + 1)), [SHAPE] * (P + 1)
    bar = threading.Barrier(world)
    slots, outs, regs, errs = [None] * world, [None] * world, [None] * world, []

    def work(rank):
        try:
            eng = ScriptedAttemptEngine(SHAPE, roiRatio, accept)
            reg = GridRegistrar(eng, roiRatio=roiRatio, directIncre=incre, window=window)
            reg.native = native
            regs[rank] = reg

            def all_gather(payload):
                slots[rank] = np.asarray(payload, np.int32)
                bar.wait()
                g = np.stack(slots)
                bar.wait()
                return g
            outs[rank] = reg.register_sharded(handles, shapes, d0, rank, world, all_gather, hint=hint)
        except BaseException as e:                            # noqa: BLE001
            errs.append(e); bar.abort()
    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ths]; [t.join() for t in ths]
    if errs:
        raise errs[0]
    return outs, regs


def test_hinted_shards_equal_sequential_whatever_the_hint():
    """A scan-pattern hint lets a rank > 0 follow ONE chain (the predicted incoming direction, predictor primed with the predicted history)
    instead of four blind ones.  Results never depend on it: right, random or partly wrong hints all give the sequential table -- a wrong
    one through the repair round (one extra all-gather, the same decision on every rank)."""
    repaired = 0
    for seed in range(25):
        rng = np.random.default_rng(4000 + seed)
        roiRatio = float(rng.choice([0.1, 0.2])); incre = int(rng.choice([-1, 1])); d0 = int(rng.integers(1, 5))
        P = int(rng.integers(6, 60)); world = int(rng.integers(2, 6)); window = int(rng.choice([3, 8, 24]))
        accept = random_truth(rng, P, roiRatio)
        seq, d_end, _ = sequential(accept, roiRatio, incre, d0)
        true_dirs = [r[3] for r in seq]
        kinds = [true_dirs, [int(rng.integers(1, 5)) for _ in range(P)], [d if rng.random() < 0.8 else d % 4 + 1 for d in true_dirs], None]
        for hint in kinds:
            outs, regs = _lockstep_sharded(accept, roiRatio, incre, d0, world, window, hint)
            for full, d in outs:
                assert d == d_end and [list(r[:4]) for r in full.tolist()] == seq, (seed, world, hint)
            assert len({getattr(r, "hint_repairs", 0) for r in regs}) == 1            # every rank took the same decision
            repaired += getattr(regs[0], "hint_repairs", 0)
    assert repaired > 5


def test_hint_removes_the_blind_start_cost_on_the_serpentine():
    """BASELINE configs[1]'s path (10 x 9 column serpentine, 89 pairs) over 8 ranks: with the scan pattern as hint the ranks together
    evaluate exactly the attempts of the sequential search (no blind starts, no slow speculation inside a chunk) and the busiest rank
    stays within 20 % of the mean -- 162 attempts, busiest 22 against a mean of 15.4 sequential, without the hint.  The native chains
    (csrc/grid.hip through vfsms_pairs_offsets_eval) take the same decisions as the Python twin."""
    accept, dirs = [], []
    for c in range(9):
        d_col = 1 if c % 2 == 0 else 3
        accept += [{(d_col, i): (3, 4) for i in range(1, 4)} for _ in range(9)]; dirs += [d_col] * 9
        if c < 8:
            accept.append({(2, i): (3, 4) for i in range(1, 4)}); dirs.append(2)
    seq, d_end, n_seq = sequential(accept, 0.2, 1, 1)
    res = {}
    for hint in (None, dirs):
        outs, regs = _lockstep_sharded(accept, 0.2, 1, 1, 8, 48, hint)
        assert all([list(r[:4]) for r in full.tolist()] == seq for full, _d in outs)
        res[hint is not None] = [r.stats["attempts"] for r in regs]
    ref = GridRegistrar(ScriptedAttemptEngine(SHAPE, 0.2, accept), roiRatio=0.2, directIncre=1, window=48)
    ref.native = False
    ref.chain(list(range(90)), [SHAPE] * 90, 0, 89, 1)
    one_gpu = ref.stats["attempts"]
    # with the prediction the ranks together evaluate exactly the attempts of the pair-by-pair search (the history-driven chain of one GPU
    # throws away two speculative ones while it learns the pattern)
    assert sum(res[True]) == n_seq <= one_gpu and max(res[True]) <= 1.2 * one_gpu / 8, (res, one_gpu, n_seq)
    assert sum(res[False]) > 1.25 * one_gpu and max(res[False]) > max(res[True])


def test_native_midpath_chain_primed_by_hint_equals_python_twin():
    """vfsms_grid_params.path_hint (csrc/grid.hip) against GridRegistrar.chain(hint=...): a chain that starts inside the path with its
    predictor primed by the predicted history evaluates the same batches in the same order in both, and the rows are the sequential
    search of the chunk -- for right and for wrong hints."""
    from imagestitch_amd._lib import pairs_offsets_eval, Engine
    for seed in range(30):
        rng = np.random.default_rng(9000 + seed)
        roiRatio = float(rng.choice([0.1, 0.2])); incre = int(rng.choice([-1, 1]))
        P = int(rng.integers(8, 70)); window = int(rng.choice([3, 8, 24, 48]))
        accept = random_truth(rng, P, roiRatio, p_fail=0.0, p_false=0.1)
        seq, _d, _n = sequential(accept, roiRatio, incre, 1)
        true_dirs = [r[3] for r in seq]
        hint = true_dirs if seed % 3 else [int(rng.integers(1, 5)) for _ in range(P)]
        lo = int(rng.integers(1, P - 2)); hi = int(rng.integers(lo + 1, P + 1))
        d_in = true_dirs[lo - 1]
        eng = ScriptedAttemptEngine(SHAPE, roiRatio, accept)
        reg = GridRegistrar(eng, roiRatio=roiRatio, directIncre=incre, window=window)
        res_py, d_py = reg.chain(list(range(P + 1)), [SHAPE] * (P + 1), lo, hi, d_in, midpath=True, hint=hint)
        log = []

        def attempts(items, accept=accept, log=log):
            rows = []
            for (k, d, i) in items:
                log.append((k, d, i))
                acc = accept[k]
                ok = (d, i) in acc
                raw = acc[(d, i)] if ok else (7, -3)
                rows.append([int(ok), raw[0], raw[1], 5 if ok else 1, 100, 100, 10, 0])
            return rows
        params = Engine.grid_params(method="surf", roiRatio=roiRatio, directIncre=incre, window=window, hint=hint)
        res_c, d_c, st = pairs_offsets_eval(attempts, [SHAPE] * (P + 1), params, lo, hi, d_in, True)
        assert np.array_equal(res_c, res_py) and d_c == d_py, seed
        assert log == eng.log and st[1] == reg.stats["batches"], (seed, len(log), len(eng.log))
        assert [list(r[:4]) for r in res_c.tolist()] == seq[lo:hi]


def test_two_process_gloo_hinted_with_repair(tmp_path):
    """world_size 2 over gloo with a scan-pattern hint that is wrong in places: the repair round (a second all-gather) runs on both ranks and
    the table is the sequential one; with the right hint there is no repair."""
    from scripted import serpentine_truth
    worker = os.path.join(os.path.dirname(__file__), "dist_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    accept = serpentine_truth(32, 32, 0.2)
    seq, d_end, _ = sequential(accept, 0.2, 1, 1)
    for mode, port in (("hint", "29623"), ("badhint", "29625")):
        out = os.path.join(str(tmp_path), "res_%s.json" % mode)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", port, worker, out, "1023", "5", "24", mode]
        subprocess.check_call(cmd, env=env, timeout=900)
        got = json.load(open(out))
        assert got["direction"] == d_end and [r[:4] for r in got["rows"]] == seq
        assert (got["repairs"] >= 1) == (mode == "badhint"), (mode, got["repairs"])


def test_serpentine_hint_on_the_first_path():
    """grid.serpentine_directions as the hint of a path registered for the FIRST time: the sequential number of attempts in three batches,
    the sequential table."""
    from imagestitch_amd.grid import serpentine_directions
    accept = []
    for c in range(9):
        d_col = 1 if c % 2 == 0 else 3
        accept += [{(d_col, i): (3, 4) for i in range(1, 4)} for _ in range(9)]
        if c < 8:
            accept.append({(2, i): (3, 4) for i in range(1, 4)})
    hint = serpentine_directions(10, 9)
    assert len(hint) == len(accept) == 89
    seq, d_end, n_seq = sequential(accept, 0.2, 1, 1)
    eng = ScriptedAttemptEngine(SHAPE, 0.2, accept)
    reg = GridRegistrar(eng, roiRatio=0.2, directIncre=1, window=48)
    res, d = reg.register(list(range(90)), [SHAPE] * 90, 1, hint=hint)
    assert [list(r[:4]) for r in res.tolist()] == seq and d == d_end
    assert reg.stats["attempts"] == n_seq and reg.stats["batches"] == 3, reg.stats


def test_path_memory_is_a_prior_never_a_result():
    """GridRegistrar.path_memory: the second registration of a scan pattern plans its batches from what the first one taught it -- the
    sequential number of attempts in a handful of batches instead of a learning phase of a dozen small ones -- and a DIFFERENT path of
    the same length registered next still comes out as its own sequential search (the stale memory only costs attempts)."""
    accept, dirs = [], []
    for c in range(9):
        d_col = 1 if c % 2 == 0 else 3
        accept += [{(d_col, i): (3, 4) for i in range(1, 4)} for _ in range(9)]
        if c < 8:
            accept.append({(2, i): (3, 4) for i in range(1, 4)})
    P = len(accept)
    seq, d_end, n_seq = sequential(accept, 0.2, 1, 1)
    for native in (False, True):
        eng = ScriptedAttemptEngine(SHAPE, 0.2, accept)
        reg = GridRegistrar(eng, roiRatio=0.2, directIncre=1, window=48)
        reg.native = native and hasattr(eng, "pairs_offsets")
        res1, d1 = reg.register(list(range(P + 1)), [SHAPE] * (P + 1), 1)
        cold = dict(reg.stats)
        res2, d2 = reg.register(list(range(P + 1)), [SHAPE] * (P + 1), 1)
        hot = {k: reg.stats[k] - cold[k] for k in cold}
        assert [list(r[:4]) for r in res1.tolist()] == seq == [list(r[:4]) for r in res2.tolist()] and d1 == d2 == d_end
        assert hot["attempts"] == n_seq <= cold["attempts"] and hot["batches"] <= 4 < cold["batches"], (cold, hot)
        # another path of the same length: the memory is wrong for it
        rng = np.random.default_rng(5)
        other = random_truth(rng, P, 0.2)
        seq_o, d_o, _n = sequential(other, 0.2, 1, 1)
        eng.accept = other
        res3, d3 = reg.register(list(range(P + 1)), [SHAPE] * (P + 1), 1)
        assert [list(r[:4]) for r in res3.tolist()] == seq_o and d3 == d_o
        reg.remember = False
        reg.register(list(range(P + 1)), [SHAPE] * (P + 1), 1)


def _serpentine_accept(rows, cols, first=1, across=2):
    """scripted truth of a column serpentine: rows - 1 pairs along `first`, one `across`, rows - 1 back, ..."""
    back = {1: 3, 3: 1, 2: 4, 4: 2}[first]
    accept = []
    for c in range(cols):
        d_col = first if c % 2 == 0 else back
        accept += [{(d_col, i): (3, 4) for i in range(1, 4)} for _ in range(rows - 1)]
        if c < cols - 1:
            accept.append({(across, i): (3, 4) for i in range(1, 4)})
    return accept


def test_a_path_memory_that_mispredicts_is_not_kept_on_trust():
    """GridRegistrar._learn: a memory is a prior on probation.  Scan patterns A (10 x 9 column serpentine) and B (the 9 x 10 one: the SAME
    number of pairs, turns elsewhere).  A, A: primed.  Then B: the stale memory costs attempts (a misprediction) -- B's pattern is adopted,
    marked suspect, and a second B is primed again (the session moved on to another pattern).  A, B, A, B on the other hand: two
    mispredictions in a row drop the memory and the next path is registered COLD, attempt for attempt and batch for batch what a fresh
    registrar does -- paths that do not repeat stop paying for each other's patterns.  Offsets always equal the sequential search."""
    A, B = _serpentine_accept(10, 9), _serpentine_accept(9, 10)
    P = len(A)
    assert len(B) == P
    seqs = {id(A): sequential(A, 0.2, 1, 1), id(B): sequential(B, 0.2, 1, 1)}

    def run(reg, eng, acc):
        eng.accept = acc
        before = dict(reg.stats)
        res, d = reg.register(list(range(P + 1)), [SHAPE] * (P + 1), 1)
        seq, d_end, _n = seqs[id(acc)]
        assert [list(r[:4]) for r in res.tolist()] == seq and d == d_end
        return {k: reg.stats[k] - before[k] for k in before}

    def cold_cost(acc):
        eng = ScriptedAttemptEngine(SHAPE, 0.2, acc)
        reg = GridRegistrar(eng, roiRatio=0.2, directIncre=1, window=48)
        return run(reg, eng, acc)
    for native in (False, True):
        eng = ScriptedAttemptEngine(SHAPE, 0.2, A)
        reg = GridRegistrar(eng, roiRatio=0.2, directIncre=1, window=48)
        reg.native = native and hasattr(eng, "pairs_offsets")
        run(reg, eng, A)
        hot = run(reg, eng, A)
        assert hot["batches"] <= 4 and not reg.path_suspect and reg.mispredictions == 0
        stale = run(reg, eng, B)                             # A's pattern predicts B badly
        assert reg.mispredictions == 1 and reg.path_suspect and reg.path_memory is not None
        assert stale["attempts"] > seqs[id(B)][2]
        again = run(reg, eng, B)                             # ... but B was learned: primed, the probation ends
        assert again["attempts"] == seqs[id(B)][2] and again["batches"] <= 4 and not reg.path_suspect and reg.mispredictions == 1
        run(reg, eng, A)                                     # B's pattern predicts A badly: suspect
        assert reg.mispredictions == 2 and reg.path_suspect
        run(reg, eng, B)                                     # second misprediction in a row: dropped
        assert reg.mispredictions == 3 and reg.path_memory is None and not reg.path_suspect
        c = run(reg, eng, A)                                 # cold, like a fresh registrar
        c0 = cold_cost(A)
        assert (c["attempts"], c["batches"]) == (c0["attempts"], c0["batches"]), (c, c0)
        assert reg.path_memory is not None and not reg.path_suspect and reg.mispredictions == 3
        # a caller's hint is never put on trial (it is the caller's to correct)
        reg.register(list(range(P + 1)), [SHAPE] * (P + 1), 1, hint=[1] * P)
        assert reg.mispredictions == 3


def test_sharded_ranks_put_a_memory_on_probation_alike():
    """The probation of a path memory (GridRegistrar._learn) is decided from the gathered table alone, so the ranks of the sharded form --
    each with a registrar of its own that lives from path to path -- must stay in step: same memory, same suspicion, same count of
    mispredictions after every path of the sequence A, A, B, A, B, A (10 x 9 and 9 x 10 serpentines: the same 89 pairs, turns elsewhere),
    and every path's table equals its sequential search whatever the memory said (repair rounds included).  Four ranks on threads with an
    in-process all-gather, Python twin and native chains."""
    import threading
    A, B = _serpentine_accept(10, 9), _serpentine_accept(9, 10)
    P, world = len(A), 4
    seqs = {id(A): sequential(A, 0.2, 1, 1), id(B): sequential(B, 0.2, 1, 1)}
    for native in (False, True):
        engs = [ScriptedAttemptEngine(SHAPE, 0.2, A) for _ in range(world)]
        regs = [GridRegistrar(e, roiRatio=0.2, directIncre=1, window=48) for e in engs]
        for r, e in zip(regs, engs):
            r.native = native and hasattr(e, "pairs_offsets")
        want_mis = 0
        for step, acc in enumerate((A, A, B, A, B, A)):
            bar = threading.Barrier(world)
            slots, outs, errs = [None] * world, [None] * world, []

            def work(rank):
                try:
                    engs[rank].accept = acc

                    def all_gather(payload):
                        slots[rank] = np.asarray(payload, np.int32)
                        bar.wait()
                        g = np.stack(slots)
                        bar.wait()
                        return g
                    outs[rank] = regs[rank].register_sharded(list(range(P + 1)), [SHAPE] * (P + 1), 1, rank, world, all_gather)
                except BaseException as e:                        # noqa: BLE001
                    errs.append(e); bar.abort()
            ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
            [t.start() for t in ths]; [t.join() for t in ths]
            if errs:
                raise errs[0]
            seq, d_end, _n = seqs[id(acc)]
            for full, d in outs:
                assert d == d_end and [list(r[:4]) for r in full.tolist()] == seq, (native, step)
            states = {(tuple(r.path_memory) if r.path_memory is not None else None, r.path_suspect, r.mispredictions) for r in regs}
            assert len(states) == 1, (native, step, states)              # every rank decided alike
            want_mis += 1 if step in (2, 3, 5) else 0                   # B on A's memory, A on B's (dropped), [B cold], A on B's
            assert regs[0].mispredictions == want_mis, (native, step, regs[0].mispredictions)
        # A, A: primed.  B: mispredicted, adopted on probation.  A: mispredicted again -> dropped.  B: cold, learned.  A: mispredicted, on probation.
        assert regs[0].path_suspect and regs[0].path_memory is not None


def test_register_projected_runs_the_ranks_own_code_one_after_the_other():
    """GridRegistrar.register_projected (bench.py --project-shards): the sharded form of N ranks in ONE process, rank after rank, through
    shard_payload / assemble / the repair round / _learn.  Its table equals the sequential search and register_sharded's, cold and primed,
    also when the memory mispredicts (paths A then B: one repair round); its per-rank attempt counts are those of ranks with registrars of
    their own (the work split the projection reports is the one an N-GPU run would have)."""
    A, B = _serpentine_accept(10, 9), _serpentine_accept(9, 10)
    P = len(A)
    seqs = {id(A): sequential(A, 0.2, 1, 1), id(B): sequential(B, 0.2, 1, 1)}
    handles, shapes = list(range(P + 1)), [SHAPE] * (P + 1)
    for world in (2, 4, 8):
        eng = ScriptedAttemptEngine(SHAPE, 0.2, A)
        reg = GridRegistrar(eng, roiRatio=0.2, directIncre=1, window=48)
        calls = []
        for acc in (A, A, B, B):
            eng.accept = acc
            primed = reg.path_memory is not None
            full, d, per_rank, tail = reg.register_projected(handles, shapes, 1, world, probe=lambda: calls.append(1) or 7.0)
            seq, d_end, n_seq = seqs[id(acc)]
            assert [list(r[:4]) for r in full.tolist()] == seq and d == d_end
            assert len(per_rank) == world and sum(q["pairs"] for q in per_rank) == P and tail >= 0.0
            assert all(q["probe"] == 7.0 and q["wall_s"] >= 0.0 for q in per_rank)
            # ranks with registrars (and engines) of their own, same memory: the same attempts per rank
            want = []
            for rank in range(world):
                e2 = ScriptedAttemptEngine(SHAPE, 0.2, acc)
                r2 = GridRegistrar(e2, roiRatio=0.2, directIncre=1, window=48)
                r2.path_memory = None if not primed else list(mem_before)
                r2.shard_payload(handles, shapes, 1, rank, world, None, r2._prediction(P, None))
                want.append(r2.stats["attempts"])
            got = [q["attempts"] for q in per_rank]
            if acc is A or not primed or mem_before == reg.path_memory:
                assert got == want, (world, got, want)
            else:                                            # B under A's memory: the first round equals, the repair round adds to the unconfirmed ranks
                assert all(g >= w for g, w in zip(got, want)) and sum(got) > sum(want) and reg.hint_repairs >= 1
            mem_before = list(reg.path_memory) if reg.path_memory is not None else None
        # primed A over 8 ranks: the sequential minimum of attempts, spread evenly
        if world == 8:
            eng.accept = A
            reg.path_memory = [int(r[3]) for r in seqs[id(A)][0]]
            full, d, per_rank, tail = reg.register_projected(handles, shapes, 1, world)
            assert sum(q["attempts"] for q in per_rank) == seqs[id(A)][2] and max(q["attempts"] for q in per_rank) <= 18


def test_probation_bookkeeping_round_6():
    """(a) a memory replaced without a trial (another path length, a caller's hint) starts clean -- it must not inherit the
    `suspect` mark of the memory it replaces; (b) the attempts a prediction promised are counted from the REAL incoming direction."""
    A, B = _serpentine_accept(10, 9), _serpentine_accept(9, 10)
    P = len(A)
    eng = ScriptedAttemptEngine(SHAPE, 0.2, A)
    reg = GridRegistrar(eng, roiRatio=0.2, directIncre=1, window=48)
    reg.register(list(range(P + 1)), [SHAPE] * (P + 1), 1)
    eng.accept = B
    reg.register(list(range(P + 1)), [SHAPE] * (P + 1), 1)
    assert reg.path_suspect                                   # B under A's memory: adopted on probation
    short = _serpentine_accept(4, 3)
    eng.accept = short
    reg.register(list(range(len(short) + 1)), [SHAPE] * (len(short) + 1), 1)     # another length: no trial, learned clean
    assert reg.path_memory is not None and len(reg.path_memory) == len(short) and not reg.path_suspect
    # (b): a path entered with direction 3 whose memory says [3, 3, ...] promised one attempt per pair
    up = [{(3, i): (3, 4) for i in range(1, 4)} for _ in range(6)]
    eng.accept = up
    reg2 = GridRegistrar(eng, roiRatio=0.2, directIncre=1, window=48)
    reg2.register(list(range(7)), [SHAPE] * 7, 3)
    reg2.register(list(range(7)), [SHAPE] * 7, 3)
    assert reg2.mispredictions == 0 and not reg2.path_suspect

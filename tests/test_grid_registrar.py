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

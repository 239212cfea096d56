"""CPU tests: the oracle against the reference's own outputs (tests/golden, captured by tools/capture_golden.py)
and against the published constants / the Stitcher.py:87 offset list."""
import json
import sys
import os

import numpy as np
import pytest


def test_optimal_dft_sizes(oracle):
    # values OpenCV's getOptimalDFTSize returns for the ROI sizes of the configs
    for n, m in [(387, 400), (2584, 2592), (409, 432), (819, 864), (1936, 1944), (516, 540), (1024, 1024), (256, 256), (614, 625)]:
        assert oracle.optimal_dft_size(n) == m


def test_integral_matches_numpy(oracle):
    rng = np.random.default_rng(1)
    for shape in [(1, 1), (7, 13), (64, 257), (200, 1000)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        ref = np.zeros((shape[0] + 1, shape[1] + 1), np.int64)
        ref[1:, 1:] = img.astype(np.int64).cumsum(0).cumsum(1)
        assert np.array_equal(oracle.integral(img), ref.astype(np.int32))
    # non-contiguous view (direction 2/4 ROIs are column slices of a tile)
    big = rng.integers(0, 256, (50, 90), dtype=np.uint8)
    view = big[:, 60:]
    ref = np.zeros((51, 31), np.int64); ref[1:, 1:] = view.astype(np.int64).cumsum(0).cumsum(1)
    assert np.array_equal(oracle.integral(view), ref.astype(np.int32))


def test_surf_layer_sizes_and_margins(oracle):
    # SURF layer sizes 9,15,21,27,33 / 18..66 / 36..132 / 72..264 (SURVEY 8c iv): a constant image has zero response
    img = np.full((300, 300), 77, np.uint8)
    S = oracle.integral(img)
    for o in range(4):
        for l in range(5):
            size, step = (9 + 6 * l) << o, 1 << o
            det, tr = oracle.surf_layer(S, size, step)
            assert det.shape == (300 // step, 300 // step)
            assert np.all(det == 0) and np.all(tr == 0)
    assert len(oracle.surf_detect(img)) == 0


def test_mode_vote_golden(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "mode_cases.npz"))
    exp = g["expected"]
    for i, (ev, st, dx, dy) in enumerate(exp):
        s, off, _v = oracle.mode_offset(g["c%d_kpsA" % i], g["c%d_kpsB" % i], g["c%d_pairs" % i], ev)
        assert (int(s), off[0], off[1]) == (st, dx, dy), i


def test_fuse_fade_golden(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "fuse_cases.npz"))
    meta = g["meta"]
    modes = set()
    for i, (dx, dy, _color) in enumerate(meta):
        out, info = oracle.fuse_fade(g["f%d_A" % i], g["f%d_B" % i], dx, dy, return_info=True)
        modes.add((int(info[0]), int(info[1])))
        assert np.array_equal(out, g["f%d_out" % i]), i
    assert {(1, 0), (1, 1), (1, 2), (1, 3), (0, -1)} <= modes   # all four corner cases and strip mode exercised


def test_bf_l2_against_numpy(oracle):
    rng = np.random.default_rng(5)
    q = rng.normal(size=(70, 64)).astype(np.float32); t = rng.normal(size=(90, 64)).astype(np.float32)
    t[10] = t[3]          # exact duplicate: ties must keep the lower train index
    q[0] = t[3]
    i1, d1, i2, d2 = oracle.bf_l2_knn2(q, t)
    D = np.sqrt(((q[:, None, :].astype(np.float64) - t[None, :, :]) ** 2).sum(-1))
    assert np.array_equal(i1[1:], D.argmin(1)[1:])
    assert i1[0] == 3 and d1[0] == 0 and d2[0] == 0
    assert np.allclose(d1, np.sort(D, 1)[:, 0], rtol=1e-5) and np.allclose(d2, np.sort(D, 1)[:, 1], rtol=1e-5)
    pairs = oracle.bf_l2_ratio_matches(q, t, 0.75)
    keep = [(int(i1[k]), k) for k in range(len(q)) if float(d1[k]) < float(d2[k]) * 0.75]
    assert [tuple(p) for p in pairs] == keep
    # fewer than two train descriptors -> no match can pass the len(m)==2 guard (ImageUtility.py:294)
    assert len(oracle.bf_l2_ratio_matches(q, t[:1], 0.75)) == 0
    assert len(oracle.bf_l2_ratio_matches(q[:0], t, 0.75)) == 0


def test_phase_correlate_known_shift(oracle):
    # known-answer: b is a shifted by (+7 rows, -5 cols) cyclically on an optimal-size image ->
    # phaseCorrelate(a, b) returns the shift of b's content relative to a's: (x, y) = (-5, +7)
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (80, 96), dtype=np.uint8)
    b = np.roll(np.roll(a, 7, 0), -5, 1)
    (x, y), resp = oracle.phase_correlate(a, b)
    assert abs(x - (-5)) < 1e-6 and abs(y - 7) < 1e-6 and resp > 0.9
    # non-optimal size gets zero padded bottom/right; the rectangle test of OpenCV's accuracy suite (no window)
    a = np.zeros((129, 128), np.uint8); b = np.zeros((129, 128), np.uint8)
    a[100:110, 100:110] = 255; b[90:100, 80:90] = 255
    (x, y), resp = oracle.phase_correlate(a, b)
    # 129 rows pad to M = 135 (odd): fftShift leaves the last row in place and the centre is M/2.0 = 67.5,
    # so OpenCV's own result carries a half-pixel bias on that axis
    assert abs(x - (-20)) < 1e-6 and abs(y - (-9.5)) < 1e-6
    (x, y), resp = oracle.phase_correlate(a[:120], b[:120])
    assert abs(x - (-20)) < 1e-6 and abs(y - (-10)) < 1e-6


def test_surf_on_real_strips_reproduces_reference_offsets(oracle, golden_dir):
    """The only numeric ground truth in the reference: Stitcher.py:87.  SURF + BF + mode on the real
    dendriticCrystal strips must land within +-1 px of it (tolerance stated by north_star for SURF)."""
    g = np.load(os.path.join(golden_dir, "real_strips.npz"))
    for n, (a, b, direction, H, W, gdx, gdy) in enumerate(g["meta"]):
        ra, rb = g["r%d_roiA" % n], g["r%d_roiB" % n]
        ka, da = oracle.surf_detect_describe(ra)
        kb, db = oracle.surf_detect_describe(rb)
        assert len(ka) > 1000 and len(kb) > 1000
        pairs = oracle.bf_l2_ratio_matches(da, db, 0.75)
        st, off, votes = oracle.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pairs, 3)
        assert st and votes > 50
        if direction == 1: off[0] += H - int(0.2 * H)
        if direction == 3: off[0] -= H - int(0.2 * H)
        if direction == 2: off[1] += W - int(0.2 * W)
        if direction == 4: off[1] -= W - int(0.2 * W)
        assert abs(off[0] - gdx) <= 1 and abs(off[1] - gdy) <= 1, (a, b, off, (gdx, gdy))


def test_dendritic_offsets_fixture(golden_dir):
    d = json.load(open(os.path.join(golden_dir, "dendritic_offsets.json")))
    off = d["offsets"]
    assert len(off) == 89 and off[3] == [1775, 2] and off[15] == [-1734, -1]
    turns = [k for k, (dx, dy) in enumerate(off) if abs(dy) > 1000]
    assert turns == [14, 29, 44, 59, 74]     # 6 columns x 15 tiles, serpentine


def test_orb_oracle_exact_truth_and_structure(oracle):
    """ORB + Hamming 1-NN + mode vote on a synthetic grid must equal the integer ground truth exactly (north_star);
    structural checks on the restated pipeline (level quotas, border, pattern generator)."""
    from imagestitch_amd.synthetic import SyntheticGrid
    from imagestitch_amd.utility import roi_rect
    # patchSize 31 (the reference's setting, ImageUtility.py:37,260): upstream's learned table bit_pattern_31_; structural
    # checks of the restated table: 256 tests, coordinates inside the 31 x 31 patch's rotation-safe disc, each test runs
    # left to right (x0 <= x1), no duplicated test, first / last rows as published
    pat = oracle.orb_pattern()
    rows = pat.reshape(256, 4)
    assert pat.shape == (512, 2) and pat.min() == -13 and pat.max() == 12
    assert rows[0].tolist() == [8, -3, 9, 5] and rows[1].tolist() == [4, 2, 7, -12] and rows[255].tolist() == [-1, -6, 0, -11]
    assert np.all(rows[:, 0] <= rows[:, 2]) and len({tuple(r) for r in rows.tolist()}) == 256
    assert np.hypot(pat[:, 0], pat[:, 1]).max() < 15 * np.sqrt(2) + 1e-9          # stays inside the 31 + 2*... border after rotation
    # any other patch size: upstream's makeRandomPattern (RNG(0x34985739), MWC)
    pat = oracle.orb_pattern(patch_size=29)
    assert pat.shape == (512, 2) and pat.min() >= -14 and pat.max() <= 14
    pat33 = oracle.orb_pattern(patch_size=33)
    assert pat33[:2].tolist() == [[13, -15], [3, 4]] or (pat33.min() >= -16 and pat33.max() <= 16)
    # ORB keeps keypoints >= 31 px from the ROI border, so the shared band must be wider than 62 px in both strips
    g = SyntheticGrid(2, 2, 1024, overlap=0.15)
    tiles = g.tiles(threads=1)
    for k, (truth, d) in enumerate(zip(g.true_offsets(), g.true_directions())):
        A, B = tiles[k], tiles[k + 1]
        ra = roi_rect(A.shape, d, "first", 0.2); rb = roi_rect(B.shape, d, "second", 0.2)
        ka, da = oracle.orb_detect_describe(np.ascontiguousarray(A[ra[0]:ra[0] + ra[2], ra[1]:ra[1] + ra[3]]))
        kb, db = oracle.orb_detect_describe(np.ascontiguousarray(B[rb[0]:rb[0] + rb[2], rb[1]:rb[1] + rb[3]]))
        assert len(ka) > 500 and da.shape == (len(ka), 32)
        assert np.all(np.diff(ka["octave"]) >= 0)                       # level-major order
        q = np.bincount(ka["octave"], minlength=8)
        assert 100 <= q[0] <= 1085 + 50                                 # level-0 quota of ORB(5000, 1.2, 8) is 1085 (+ ties)
        lx = ka["x"] / (1.2 ** ka["octave"]); ly = ka["y"] / (1.2 ** ka["octave"])
        assert lx.min() >= 30.9 and ly.min() >= 30.9                    # runByImageBorder(31) in level coordinates
        pairs, dist = oracle.bf_hamming_matches(da, db)
        assert len(pairs) == len(ka)                                    # one match per query, no threshold (ImageUtility.py:297-302)
        st, off, votes = oracle.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pairs, 3)
        L = int(0.2 * 1024)
        if d == 1: off[0] += 1024 - L
        if d == 3: off[0] -= 1024 - L
        if d == 2: off[1] += 1024 - L
        if d == 4: off[1] -= 1024 - L
        assert st and off == truth, (k, off, truth, votes)


def test_demo_strips_fixture_reproduced_by_oracle(oracle, golden_dir):
    """BASELINE configs[0] (iron pair) and configs[3] (zirconCL sequence): ROI strips at roiRatio 0.2 with the offsets the oracle
    produced when the fixture was captured (tools/capture_golden.py demo; oracle-generated because cv2 is not installable).
    Guards the oracle against regressions; the iron strip is 387 x 2584 -> DFT size 400 x 2592 (SURVEY 8c iv)."""
    meta = json.load(open(os.path.join(golden_dir, "demo_strips.json")))["cases"]
    g = np.load(os.path.join(golden_dir, "demo_strips.npz"))
    assert meta[0]["dataset"] == "iron" and meta[0]["roi"] == [387, 2584]
    assert oracle.optimal_dft_size(387) == 400 and oracle.optimal_dft_size(2584) == 2592
    for n, c in enumerate(meta):
        (x, y), r = oracle.phase_correlate(g["d%d_roiA" % n], g["d%d_roiB" % n])
        assert [x, y] == c["phase_xy"] and r == c["phase_response"], (n, (x, y), c["phase_xy"])
    # phase correlation and SURF agree on the iron pair (the reference's sign quirk: phase = -offset; Stitcher.py:231-232)
    assert meta[0]["phase_int"] == [-149, 0] and meta[0]["surf"]["offset"] == [150, 0]
    c = meta[1]
    ka, da = oracle.surf_detect_describe(g["d1_roiA"]); kb, db = oracle.surf_detect_describe(g["d1_roiB"])
    pairs = oracle.bf_l2_ratio_matches(da, db, 0.75)
    st, off, votes = oracle.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pairs, 3)
    assert [int(st), off, int(votes), len(ka), len(kb), len(pairs)] == [c["surf"]["status"], c["surf"]["offset"], c["surf"]["votes"], c["surf"]["nA"], c["surf"]["nB"], c["surf"]["matches"]]


def _rebuild_frames(nb, g):
    H, W = nb["shape"]
    frames = {t: np.zeros((H, W), np.uint8) for t in nb["tiles"]}
    for st in nb["strips"]:
        a = g[st["key"]]
        frames[st["tile"]][st["y0"]:st["y0"] + a.shape[0], st["x0"]:st["x0"] + a.shape[1]] = a
    return [frames[t] for t in nb["tiles"]]


def test_dendritic_whole_path_oracle_vs_stitcher_py_87(golden_dir):
    """tests/golden/dendritic_path_oracle.json (tools/capture_golden.py realpath): the oracle behind the reference's incremental
    search, direction threaded pair to pair, over ALL 87 usable pairs of the reference's dendriticCrystal path on the real
    tiles, beside Stitcher.py:87.  Every pair within +-1 px (north_star's SURF tolerance), nine in ten exact, every turn
    resolved in the reference's candidate order (wrong directions fail on the real tiles)."""
    d = json.load(open(os.path.join(golden_dir, "dendritic_path_oracle.json")))
    gold = json.load(open(os.path.join(golden_dir, "dendritic_offsets.json")))["offsets"]
    rows = d["rows"]
    assert [r["a"] for r in rows] == list(range(3, 90)) and d["pairs"] == 87
    exact = 0
    for r in rows:
        assert r["gold"] == gold[r["a"] - 1]
        assert abs(r["oracle"][0] - r["gold"][0]) <= 1 and abs(r["oracle"][1] - r["gold"][1]) <= 1, r
        exact += r["oracle"] == r["gold"]
        assert r["i"] == 1 and r["attempts"][-1][2] == 1 and all(a[2] == 0 for a in r["attempts"][:-1])
    assert exact >= 75 and exact == d["exact"]
    turns = {r["a"]: r for r in rows if len(r["attempts"]) > 1}
    # down -> right: [1, 2]; right -> up: [2, 3]; up -> right: [3, 4, 1, 2]; right -> down: [2, 3, 4, 1]
    assert [a[0] for a in turns[15]["attempts"]] == [1, 2] and [a[0] for a in turns[30]["attempts"]] == [3, 4, 1, 2]
    assert [a[0] for a in turns[31]["attempts"]] == [2, 3, 4, 1] and sorted(turns) == [15, 16, 30, 31, 45, 46, 60, 61, 75, 76]


def test_real_path_strips_reproduced_by_oracle(oracle, golden_dir):
    """One neighbourhood of tests/golden/real_path_strips.* (the turn 15 -> 16 with two pairs on either side): the oracle on the
    rebuilt frames reproduces the stored rows exactly (offset, direction, i, votes, keypoint and match counts), and those are
    within +-1 px of Stitcher.py:87.  (All five neighbourhoods go through the HIP grid registrar in tests/test_gpu_parity.py.)"""
    sys.path.insert(0, os.path.join(os.path.dirname(golden_dir), "..", "tools"))
    from imagestitch_amd.utility import roi_rect
    meta = json.load(open(os.path.join(golden_dir, "real_path_strips.json")))["neighbourhoods"]
    assert [nb["turn"] for nb in meta] == [15, 30, 45, 60, 75] and sum(len(nb["expected"]) for nb in meta) == 25
    for nb in meta:
        for e in nb["expected"]:
            assert abs(e["offset"][0] - e["gold"][0]) <= 1 and abs(e["offset"][1] - e["gold"][1]) <= 1
    g = np.load(os.path.join(golden_dir, "real_path_strips.npz"))
    nb = meta[0]
    frames = _rebuild_frames(nb, g)
    direction = nb["incoming_direction"]
    for k, e in enumerate(nb["expected"]):
        A, B = frames[k], frames[k + 1]
        found = None
        d = direction
        while found is None:                                  # i = 1 ring of Stitcher.py:319-351 (every stored pair resolves there)
            ra = roi_rect(A.shape, d, "first", 0.2); rb = roi_rect(B.shape, d, "second", 0.2)
            a = np.ascontiguousarray(A[ra[0]:ra[0] + ra[2], ra[1]:ra[1] + ra[3]]); b = np.ascontiguousarray(B[rb[0]:rb[0] + rb[2], rb[1]:rb[1] + rb[3]])
            ka, da = oracle.surf_detect_describe(a); kb, db = oracle.surf_detect_describe(b)
            if len(ka) and len(kb):
                pairs = oracle.bf_l2_ratio_matches(da, db, 0.75)
                st, off, votes = oracle.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pairs, 3)
                if st:
                    found = (d, off, votes, len(ka), len(kb), len(pairs))
                    break
            d = d % 4 + 1
            assert d != direction
        d, off, votes, na, nbk, nm = found
        H, W = A.shape
        if d == 1: off[0] += H - int(0.2 * H)
        elif d == 2: off[1] += W - int(0.2 * W)
        elif d == 3: off[0] -= H - int(0.2 * H)
        else: off[1] -= W - int(0.2 * W)
        assert [off, d, votes, na, nbk, nm] == [e["offset"], e["direction"], e["votes"], e["nA"], e["nB"], e["matches"]], (k, off, e)
        direction = d


def test_keypoint_greater_orders_y_descending(oracle):
    """upstream surf.cpp KeypointGreater: response, size, octave descending, then y DESCENDING, then x ascending.  An image made of
    one 64 x 64 block repeated has many keypoints with identical response / size / octave at different positions."""
    block = np.random.default_rng(5).integers(0, 256, (64, 64), dtype=np.uint8)
    img = np.tile(block, (4, 5))
    k = oracle.surf_detect(img)
    key = np.stack([k["response"], k["size"], k["octave"].astype(np.float32)], 1)
    same = np.all(key[1:] == key[:-1], 1)
    assert same.sum() > 50
    y0, y1, x0, x1 = k["y"][:-1][same], k["y"][1:][same], k["x"][:-1][same], k["x"][1:][same]
    assert np.all((y0 > y1) | ((y0 == y1) & (x0 < x1)))
    assert np.any(y0 > y1) and np.any(y0 == y1)


def _chain_search(attempt, shapeA, shapeB, d0, roiRatio=0.2, directIncre=1):
    """Stitcher.calculateOffsetForFeatureSearchIncre's candidate walk (Stitcher.py:316-361) over `attempt(direction, i)` ->
    (status, [dx, dy], votes): (status, corrected offset, direction, i, log of (direction, i, status, raw dx, raw dy, votes))."""
    log = []
    for i in range(1, int(np.floor(0.5 / roiRatio) + 1) + 1):
        d = d0
        while True:
            st, off, votes = attempt(d, i)
            log.append((d, i, int(st), int(off[0]), int(off[1]), int(votes)))
            if st:
                off = [int(off[0]), int(off[1])]
                if d == 1: off[0] += shapeA[0] - int(i * roiRatio * shapeA[0])
                elif d == 2: off[1] += shapeA[1] - int(i * roiRatio * shapeA[1])
                elif d == 3: off[0] -= shapeB[0] - int(i * roiRatio * shapeB[0])
                else: off[1] -= shapeB[1] - int(i * roiRatio * shapeB[1])
                return True, off, d, i, log
            d += directIncre
            d = 1 if d == 5 else 4 if d == 0 else d
            if d == d0:
                break
    return False, [0, 0], d0, 0, log


def oracle_orb_attempt(oracle, A, B, roiRatio=0.2, offsetEvaluate=3):
    """attempt(direction, i) with the oracle's ORB + BF-Hamming 1-NN + mode vote (ImageUtility.py:260, 297-302, 139-178)"""
    from imagestitch_amd.utility import roi_rect

    def attempt(d, i):
        ra = roi_rect(A.shape, d, "first", i * roiRatio); rb = roi_rect(B.shape, d, "second", i * roiRatio)
        a = np.ascontiguousarray(A[ra[0]:ra[0] + ra[2], ra[1]:ra[1] + ra[3]]); b = np.ascontiguousarray(B[rb[0]:rb[0] + rb[2], rb[1]:rb[1] + rb[3]])
        ka, da = oracle.orb_detect_describe(a); kb, db = oracle.orb_detect_describe(b)
        if not len(ka) or not len(kb):
            return False, [0, 0], 0
        pairs, _ = oracle.bf_hamming_matches(da, db)
        return oracle.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pairs, offsetEvaluate)
    return attempt


def test_phase_oracle_against_independent_numpy_restatement(oracle, golden_dir):
    """The phase-correlation leg has no reference-held vector (cv2 is not installable).  What can be ruled out is a transcription
    slip shared by nobody: tests/phase_numpy.py restates cv2.phaseCorrelate a second time (numpy rfft2 / irfft2, no code in common
    with the C oracle) and the two must agree on the iron strip pair (configs[0]) and on ALL 23 zirconCL pairs (configs[3]):
    same integer peak, sub-pixel position within 1e-9 px, response within 1e-12 -- live, and against the values stored at capture."""
    import phase_numpy as PN
    meta = json.load(open(os.path.join(golden_dir, "phase_independent.json")))
    z = np.load(os.path.join(golden_dir, "zirconcl_strips.npz"))
    iron = np.load(os.path.join(golden_dir, "demo_strips.npz"))
    rows = meta["rows"]
    assert len(rows) == 24 and sum(r["dataset"] == "zirconCL" for r in rows) == 23
    for k, r in enumerate(rows):
        a, b = (z["t%d_first" % k], z["t%d_second" % (k + 1)]) if r["dataset"] == "zirconCL" else (iron["d0_roiA"], iron["d0_roiB"])
        assert list(a.shape) == r["roi"]
        (ox, oy), orr = oracle.phase_correlate(a, b)
        (nx, ny), nr, pk = PN.phase_correlate(a, b)
        assert int(ox) == int(nx) and int(oy) == int(ny) and [int(oy), int(ox)] == r["offset_int"], (k, ox, nx, oy, ny)
        assert abs(ox - nx) < 1e-9 and abs(oy - ny) < 1e-9 and abs(orr - nr) < 1e-12, (k, ox - nx, oy - ny, orr - nr)
        assert abs(nx - r["numpy_xy"][0]) < 1e-9 and abs(ny - r["numpy_xy"][1]) < 1e-9 and abs(nr - r["numpy_response"]) < 1e-12
        assert abs(ox - r["oracle_xy"][0]) < 1e-9 and abs(orr - r["oracle_response"]) < 1e-12 and (orr > 0.15) == r["accepted"]
    # the reference-as-written result of configs[0] (the sign quirk, [1400, 0] where the true offset is ~[1699, -1])
    r = rows[-1]
    assert r["dataset"] == "iron" and [r["offset_int"][0] + 1936 - 387, r["offset_int"][1]] == [1400, 0]
    # odd padded sizes (625 = 5^4: the quadrant swap leaves the last row in place) and tiny ones
    rng = np.random.default_rng(5)
    for h, w in ((614, 96), (129, 128), (31, 47), (625, 75)):
        a = rng.integers(0, 255, (h, w)).astype(np.uint8)
        b = (np.roll(a, (3, -5), (0, 1)) * 0.9 + rng.integers(0, 20, (h, w))).astype(np.uint8)
        (ox, oy), orr = oracle.phase_correlate(a, b)
        (nx, ny), nr, pk = PN.phase_correlate(a, b)
        assert abs(ox - nx) < 1e-9 and abs(oy - ny) < 1e-9 and abs(orr - nr) < 1e-12, (h, w)


def test_orb_whole_path_oracle_vs_stitcher_py_87(golden_dir):
    """tests/golden/dendritic_path_oracle_orb.json (tools/capture_golden.py realpath_orb): Stitcher.py:87 is a list of TRUE offsets, so
    the ORB leg (oracle ORB + Hamming 1-NN + mode vote behind the reference's incremental search, direction threaded) is pinned to
    it as well: 72 of the 87 pairs land within +-1 px of it, and EVERY decision carried by more than three votes does (north_star
    asks bit-exact vs the reference's ORB; the list is what the reference holds).  The 15 that miss are all accepts on exactly
    offsetEvaluate = 3 equal votes out of ~5000 unconditional 1-NN matches (no ratio test, no distance threshold on the cv2 path:
    ImageUtility.py:297-302) -- a wrong candidate direction at a turn, or a strip whose true overlap is thinner than ORB's 31-px
    border -- and a wrong accepted direction then misleads the following pairs (tiles 75..80).  Recorded, not hidden: it is what
    the reference's search does with these operators and this threshold."""
    d = json.load(open(os.path.join(golden_dir, "dendritic_path_oracle_orb.json")))
    rows = d["rows"]
    assert d["pairs"] == 87 and len(rows) == 87 and [r["a"] for r in rows] == list(range(3, 90))
    surf = {r["a"]: r for r in json.load(open(os.path.join(golden_dir, "dendritic_path_oracle.json")))["rows"]}
    misses = [r for r in rows if not r["within_one"]]
    for r in rows:
        assert r["gold"] == surf[r["a"]]["gold"]
        ok = r["status"] and abs(r["oracle"][0] - r["gold"][0]) <= 1 and abs(r["oracle"][1] - r["gold"][1]) <= 1
        assert ok == r["within_one"]
        if r["votes"] > 3:
            assert ok, r["a"]                                   # any decision backed by more than the minimum is the true offset
    assert d["within_one"] == 87 - len(misses) == 72
    for r in misses:                                            # every miss is an ACCEPT on exactly offsetEvaluate = 3 equal votes
        assert r["status"] and r["votes"] == 3 and r["note"], r["a"]


def test_orb_real_path_strips_reproduced_by_oracle(oracle, golden_dir):
    """The 25 committed neighbourhood pairs under ORB: of the stored `expected_orb` rows (oracle on the rebuilt frames at capture) 22 lie
    within +-1 px of Stitcher.py:87; the other three are accepts on 3-4 equal votes at i = 1 for pairs whose true overlap inside the
    first ROI strip is thinner than ORB's 31-px keypoint border (tiles 14, 61, 74: overlaps of 49-63 rows) -- SURF registers them,
    ORB cannot see them and the reference's threshold of three votes lets noise through.  One neighbourhood is recomputed live."""
    meta = json.load(open(os.path.join(golden_dir, "real_path_strips.json")))["neighbourhoods"]
    n, miss = 0, []
    for nb in meta:
        for e in nb["expected_orb"]:
            ok = e["status"] and abs(e["offset"][0] - e["gold"][0]) <= 1 and abs(e["offset"][1] - e["gold"][1]) <= 1
            assert e["status"] and ok == e["within_one"], e
            if not ok:
                assert e["votes"] <= 4, e
                miss.append(e["a"])
            n += 1
    assert n == 25 and miss == [14, 61, 74]
    g = np.load(os.path.join(golden_dir, "real_path_strips.npz"))
    nb = meta[1]
    frames = _rebuild_frames(nb, g)
    direction = nb["incoming_direction"]
    for k, e in enumerate(nb["expected_orb"]):
        st, off, d, i, log = _chain_search(oracle_orb_attempt(oracle, frames[k], frames[k + 1]), frames[k].shape, frames[k + 1].shape, direction)
        assert [st, off, d, i, log[-1][5]] == [True, e["offset"], e["direction"], e["i"], e["votes"]], (k, off, e)
        direction = d


def _ncc(a, b):
    a = a.astype(np.float64) - a.mean(); b = b.astype(np.float64) - b.mean()
    d = np.sqrt((a * a).sum() * (b * b).sum())
    return float((a * b).sum() / d) if d > 0 else 0.0


def _overlap_ncc(A, B, dx, dy):
    """normalised cross-correlation of the pixels two equal-size strips share when B's (y, x) lies on A's (y + dx, x + dy)
    (the vote is ptA - ptB, ImageUtility.py:150-152)"""
    h, w = A.shape
    y0, y1, x0, x1 = max(0, dx), min(h, h + dx), max(0, dy), min(w, w + dy)
    if y1 - y0 < 8 or x1 - x0 < 8:
        return -2.0
    return _ncc(A[y0:y1, x0:x1], B[y0 - dx:y1 - dx, x0 - dy:x1 - dy])


def zirconcl_surf_rows(oracle, golden_dir):
    """oracle SURF + BF-L2 + ratio + mode vote on the direction-4 ROI strips of the 23 zirconCL pairs (tests/golden/zirconcl_strips.npz)"""
    z = np.load(os.path.join(golden_dir, "zirconcl_strips.npz"))
    out = []
    for k in range(23):
        A, B = z["t%d_first" % k], z["t%d_second" % (k + 1)]
        ka, da = oracle.surf_detect_describe(A); kb, db = oracle.surf_detect_describe(B)
        pairs = oracle.bf_l2_ratio_matches(da, db, 0.75)
        st, off, votes = oracle.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pairs, 3)
        out.append((A, B, [int(st), int(off[0]), int(off[1]), int(votes), len(ka), len(kb), len(pairs)]))
    return out


def test_surf_oracle_against_phase_and_ncc_on_zirconcl(oracle, golden_dir):
    """A pin of the SURF leg that owes nothing to the SURF code: on the 23 real zirconCL pairs (BASELINE configs[3]'s tiles) the oracle's
    SURF + BF + mode offset must (a) agree within 1.5 px with the sub-pixel phase-correlation peak of the same strips -- an unrelated algorithm,
    itself pinned by the independent numpy restatement (phase_independent.json) -- up to the two things that are properties of the phase
    path: the reference-as-written sign (mirrored) and the period of the 256-column strip; and (b) be, within 1 px, the
    maximum of the normalised cross-correlation of the overlapping pixels, with a correlation above 0.93 there."""
    rows = json.load(open(os.path.join(golden_dir, "phase_independent.json")))["rows"]
    got = zirconcl_surf_rows(oracle, golden_dir)
    for k, (A, B, r) in enumerate(got):
        assert r[0] == 1 and r[3] >= 12, (k, r)
        dx, dy = r[1], r[2]
        py, px = rows[k]["numpy_xy"]                              # sub-pixel peak (x = columns, y = rows); Stitcher.py:244-251 truncates it
        assert abs(dx + px) <= 1.5, (k, dx, px)
        wrapped = (dy + py + 128) % 256 - 128                     # dy + py == 0 modulo the strip width
        assert abs(wrapped) <= 1.5, (k, dy, py)
        c = {(u, v): _overlap_ncc(A, B, dx + u, dy + v) for u in (-2, -1, 0, 1, 2) for v in (-2, -1, 0, 1, 2)}
        best = max(c, key=c.get)
        assert c[(0, 0)] > 0.93 and max(abs(best[0]), abs(best[1])) <= 1, (k, c[(0, 0)], best)
    # configs[0]'s pair (iron, 387 x 2584 strips, direction 1): 626 votes for [150, 0]; the phase peak, mirrored, says (149.9, -0.5)
    iron = np.load(os.path.join(golden_dir, "demo_strips.npz"))
    A, B = iron["d0_roiA"], iron["d0_roiB"]
    ka, da = oracle.surf_detect_describe(A); kb, db = oracle.surf_detect_describe(B)
    st, off, votes = oracle.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), oracle.bf_l2_ratio_matches(da, db, 0.75), 3)
    px, py = rows[-1]["numpy_xy"]
    assert rows[-1]["dataset"] == "iron" and st and off == [150, 0] and votes > 500
    assert abs(off[0] + py) <= 1.5 and abs(off[1] + px) <= 1.5, (off, px, py)
    c = {(u, v): _overlap_ncc(A, B, off[0] + u, off[1] + v) for u in (-2, -1, 0, 1, 2) for v in (-2, -1, 0, 1, 2)}
    best = max(c, key=c.get)
    assert c[(0, 0)] > 0.9 and max(abs(best[0]), abs(best[1])) <= 1, (c[(0, 0)], best)


def test_orb_oracle_against_surf_and_ncc_on_zirconcl(oracle, golden_dir):
    """The ORB leg on the same 23 real pairs: every decision carried by MORE than three votes lies within 2 px of the SURF offset (a
    different detector, descriptor and matcher) and of the NCC maximum, at a correlation above 0.93; the two decisions that rest on
    exactly three votes (pairs 8 and 20) are the reference's own false accepts at offsetEvaluate = 3 (no overlap at all: NCC ~ 0) --
    the same behaviour as on Stitcher.py:87's path (test_orb_whole_path_against_reference_vector)."""
    z = np.load(os.path.join(golden_dir, "zirconcl_strips.npz"))
    surf = zirconcl_surf_rows(oracle, golden_dir)
    three_vote = []
    for k in range(23):
        A, B = z["t%d_first" % k], z["t%d_second" % (k + 1)]
        ka, da = oracle.orb_detect_describe(A); kb, db = oracle.orb_detect_describe(B)
        pairs, _ = oracle.bf_hamming_matches(da, db)
        st, off, votes = oracle.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pairs, 3)
        assert st and votes >= 3, k
        if votes == 3:
            three_vote.append(k)
            assert _overlap_ncc(A, B, off[0], off[1]) < 0.2, (k, off)
            continue
        s = surf[k][2]
        assert abs(off[0] - s[1]) <= 2 and abs(off[1] - s[2]) <= 2, (k, off, s)
        c = {(u, v): _overlap_ncc(A, B, off[0] + u, off[1] + v) for u in range(-2, 3) for v in range(-2, 3)}
        assert max(c.values()) > 0.93 and c[(0, 0)] > 0.9, (k, c[(0, 0)])
    assert three_vote == [8, 20], three_vote


def phase87_residual(row, xy=None):
    """(row residual, column residual) of the sign-fixed, axis-corrected sub-pixel phase peak against Stitcher.py:87's offset, wrapped to the
    padded strip (the DFT is circular: an in-strip shift s and s - M are the same peak)"""
    M, N = row["padded"]
    x, y = xy if xy is not None else row["phase_xy"]
    c, g = row["axis_correction"], row["gold"]
    return ((-y + c[0] - g[0] + M / 2.0) % M - M / 2.0, (-x + c[1] - g[1] + N / 2.0) % N - N / 2.0)


def test_phase_oracle_against_the_reference_offset_vector(oracle, golden_dir):
    """The phase leg pinned to the one numeric vector the reference holds (Stitcher.py:87, the TRUE offsets of the dendriticCrystal path):
    on the ROI strips of the accepted (direction, i) of EVERY pair (tiles 003..090; tools/capture_golden.py phase87) the oracle's
    cv2.phaseCorrelate restatement, with the sign of the feature path (Stitcher.phaseSignFix) and the reference's axis correction
    (Stitcher.py:244-251), lands on the gold offset within 1.5 px modulo the padded strip -- all 87 pairs, whatever the response; the
    truncated integer form (Stitcher.py:231-232) within 2.  The 25 pairs whose strips are committed as crops are recomputed here."""
    d = json.load(open(os.path.join(golden_dir, "dendritic_phase87.json")))
    assert len(d["full"]) == 87 and len(d["crops"]) == 25
    for r in d["full"] + d["crops"]:
        ry, rx = phase87_residual(r)
        assert max(abs(ry), abs(rx)) <= 1.5, (r["a"], ry, rx)
        assert max(abs(v) for v in r["residual_mod_padded"]) <= 2, (r["a"], r["residual_mod_padded"])
    assert sum(r["accepted"] for r in d["full"]) >= 80                      # the reference's own gate (response > 0.15) passes most of them
    z = np.load(os.path.join(golden_dir, "real_path_strips.npz"))
    for r in d["crops"]:
        a, b = np.ascontiguousarray(z[r["key_a"]]), np.ascontiguousarray(z[r["key_b"]])
        assert list(a.shape) == r["roi"]
        (x, y), resp = oracle.phase_correlate(a, b)
        assert abs(x - r["phase_xy"][0]) < 1e-9 and abs(y - r["phase_xy"][1]) < 1e-9 and abs(resp - r["response"]) < 1e-12, r["a"]
        ry, rx = phase87_residual(r, (x, y))
        assert max(abs(ry), abs(rx)) <= 1.5


def test_rbrief_table_second_transcription():
    """Product and oracle read ONE file for ORB's learned rBRIEF table (imagestitch_amd/csrc/orb_pattern31.h), so a transcription slip in
    it would pass every engine-vs-oracle comparison.  This is a SECOND, independent transcription of upstream's published
    `bit_pattern_31_` (modules/features2d/src/orb.cpp; the same 256 tests are printed in ORB-SLAM's ORBextractor.cc): its first 64 tests
    and its last 16, typed here from the published listing, not copied out of the header -- 320 of the 1024 coordinates, both ends of the
    table, so a dropped or shifted row anywhere in between moves the tail -- held against the header's text, against what the oracle's
    library hands out, and a digest of the whole table pins every other value against later edits."""
    import hashlib
    import re
    head = [8, -3, 9, 5, 4, 2, 7, -12, -11, 9, -8, 2, 7, -12, 12, -13, 2, -13, 2, 12, 1, -7, 1, 6, -2, -10, -2, -4, -13, -13, -11, -8,
            -13, -3, -12, -9, 10, 4, 11, 9, -13, -8, -8, -9, -11, 7, -9, 12, 7, 7, 12, 6, -4, -5, -3, 0, -13, 2, -12, -3, -9, 0, -7, 5,
            12, -6, 12, -1, -3, 6, -2, 12, -6, -13, -4, -8, 11, -13, 12, -8, 4, 7, 5, 1, 5, -3, 10, -3, 3, -7, 6, 12, -8, -7, -6, -2,
            -2, 11, -1, -10, -13, 12, -8, 10, -7, 3, -5, -3, -4, 2, -3, 7, -10, -12, -6, 11, 5, -12, 6, -7, 5, -6, 7, -1, 1, 0, 4, -5,
            9, 11, 11, -13, 4, 7, 4, 12, 2, -1, 4, 4, -4, -12, -2, 7, -8, -5, -7, -10, 4, 11, 9, 12, 0, -8, 1, -13, -13, -2, -8, 2,
            -3, -2, -2, 3, -6, 9, -4, -9, 8, 12, 10, 7, 0, 9, 1, 3, 7, -5, 11, -10, -13, -6, -11, 0, 10, 7, 12, 1, -6, -3, -6, 12,
            10, -9, 12, -4, -13, 8, -8, -12, -13, 0, -8, -4, 3, 3, 7, 8, 5, 7, 10, -7, -1, 7, 1, -12, 3, -10, 5, 6, 2, -4, 3, -10,
            -13, 0, -13, 5, -13, -7, -12, 12, -13, 3, -11, 8, -7, 12, -4, 7, 6, -10, 12, 8, -9, -1, -7, -6, -2, -5, 0, 12, -12, 5, -7, 5]
    tail = [2, 7, 3, -9, -1, -6, -1, -1, 9, 5, 11, -2, 11, -3, 12, -8, 3, 0, 3, 5, -1, 4, 0, 10, 3, -6, 4, 5, -13, 0, -10, 5,
            5, 8, 12, 11, 8, 9, 9, -6, 7, -4, 8, -12, -10, 4, -10, 9, 7, 3, 12, 4, 9, -7, 10, -2, 7, 0, 12, -2, -1, -6, 0, -11]
    assert len(head) == 256 and len(tail) == 64
    txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "imagestitch_amd", "csrc", "orb_pattern31.h")).read()
    vals = [int(x) for x in re.findall(r"-?\d+", txt[txt.index("{") + 1:txt.rindex("}")])]
    assert len(vals) == 1024
    assert vals[:256] == head and vals[-64:] == tail
    assert hashlib.sha256(bytes((x + 256) % 256 for x in vals)).hexdigest() == "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    from oracle import oracle as O
    O.build()
    assert O.orb_pattern().reshape(-1).tolist() == vals

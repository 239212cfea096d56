"""CPU tests: the oracle against the reference's own outputs (tests/golden, captured by tools/capture_golden.py)
and against the published constants / the Stitcher.py:87 offset list."""
import json
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
    pat = oracle.orb_pattern()
    assert pat.shape == (512, 2) and pat.min() >= -15 and pat.max() <= 15 and pat[:2].tolist() == [[13, -15], [3, 4]]
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

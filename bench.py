#!/usr/bin/env python
"""bench.py -- image-pairs/sec of the VFSMS registration hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric / SURVEY section 8d): a synthetic 10 x 9 grid of 2048 x 2048 grayscale tiles on a
column-major serpentine path (89 consecutive pairs, 8 turns), SURF (hessian 100, 4 octaves, 3 layers, 64-d) +
BF-L2 2-NN + ratio 0.75 + mode vote (>= 3), incremental ROI (roiRatio 0.2) with direction rotation
(direction 1, directIncre 1) -- exactly Main.py's settings.  One "step" = registering all 89 pairs.

`value`: tiles already resident in HBM when the timed region starts.  `value_host_resident_tiles`: the same K steps with
the tiles in (pinned) host memory at the start of every step -- uploaded inside the timed region on the engine's copy
stream, overlapped with the registration of earlier pairs (SURVEY 8d's reading: "tiles decoded and resident in host memory").

With N > 1 the pairs are sharded in contiguous chunks, one process per GPU, and ONE all-gather (RCCL) of the int32 offset
tables closes the step ("strong" scaling: total work is fixed).  Prints ONE JSON line on rank 0.

Path memory (round 4): the registrar remembers the accepted directions of the path it registered last and plans the speculative batches of
the next path of the same length from them (GridRegistrar.path_memory; a prior like a branch predictor's: every attempt is still evaluated,
results never depend on it).  The warm-up step teaches it the grid's scan pattern, the K timed steps run on it: 3 batches per step instead
of 13, one primed chain per rank at N > 1.  `value_cold_path` (N = 1) is the same K steps by a registrar without memory -- the
configuration of rounds 1-3; --no-path-memory makes it the headline.

The prior of the timed steps is learned on ANOTHER instance of the scan pattern (--prior other, the default since round 5: same geometry,
seed + 1 -- other texture, jitter and offsets --, registered once, cold, before the warm-up: what the second dataset of a session has;
--prior same: only the warm-up steps of the timed grid teach it, round 4's headline).

The surf roofline counts the REFERENCE's 26 lane-operations per bilinear sample (DESC_OPS_LOWER_BOUND) over the live HIP-event duration of the
descriptor stage, against the calibrated VALU peak at 2.4 GHz (`frac`) and at the clock the kernel runs at (`frac_at_effective_clock`); its
PMC-derived fields come from the newest profiles/*_pmc_summary.txt, and `pmc.pmc_stale` says whether that summary was taken of the kernel
sources this run loaded (tools/build_id.py).

--method orb | phase | fuse time the other paths of the scope table on the same grid (each with its own roofline object);
the default, surf, is the BASELINE metric.  --rows 32 --cols 32 --tile 4096 [--also-fuse] is BASELINE configs[4] on one GPU (tiles synthesised
by worker processes; --also-fuse prints the mosaic-assembly line of the same resident tiles as a second JSON line).  --force-dist (N = 1):
the step's all-gather through torch.distributed "nccl" (RCCL) at world size 1.
--workload dendritic25 (N = 1): the 25 committed pairs of the reference's dendriticCrystal set (tests/golden/real_path_strips.*), a second
SURF line on real texture.  --from-files (N = 1): the same grid as JPEG files through Stitcher's ingest pipeline, decode inclusive.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak of gfx950 (v_mfma_f32_32x32x16_bf16), MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3       # dense FP32 MFMA peak of gfx950 (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md
# VALU issue peak in lane-operations: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T/s, i.e. one wave64 VALU instruction occupies its SIMD
# for 4 cycles.  CALIBRATED on the box (tools/valu_peak.hip, every SIMD filled with 1-8 waves of independent chains,
# profiles/r05_valu_peak.txt): v_fma_f32 38.3, v_fma_f64 33.2, v_add_f64 36.9, v_cvt_f32_ubyte0 37.8, v_cvt_i32_f64 37.5, v_cvt_f32_f64 36.8,
# v_pk_mul_f32 33.1 T lane-ops/s -- the 4-cycle rate (0.84-0.97 of 39.3) for every VOP3 / conversion / f64 / packed instruction the
# descriptor kernels lean on; MI355X_MICROARCH.md's "SIMD-32, 2 cycles" (78.6 T/s) is NOT reached by them.  Only plain VOP2 operations run
# faster: v_mul_f32 56.4, v_add_u32 60.7 T/s (~2.7 cycles).  157.3 TFLOP/s = 39.3 T x 2 flop per FMA x 2 for packed FP32.
VALU_PEAK_TLANEOPS = 256 * 4 * 16 * 2.4e9 / 1e12
# VALU instructions per bilinear sample of k_describe's staging loop, counted in its gfx950 assembly (stage_round<4>: 99 per four samples of a
# lane -- cvt_f64_i32 + 2 v_fma_f64 + 6 v_add_f64 for the positions, 8 v_cvt_i32_f64, 4 v_mul_u32_u24 + 4 v_add_lshl for the addresses, 8
# v_fract_f64, 8 v_cvt_f32_f64, 8 v_sub, 16 v_cvt_f32_ubyte, 16 v_pk_mul, 16 v_add, 2 address / counter adds); reported beside the roofline.
# The roofline itself counts the REFERENCE's 26 operations per sample (below): a number that does not move when the kernel is rewritten.
DESC_VALU_PER_SAMPLE = 24.75
# An op count that does not depend on how the kernel is written: the arithmetic the REFERENCE's expression needs per window sample
# (SURFInvoker: pixel_x += cos, pixel_y -= sin (2), two floors (2), two (float)(p - i) (2), four u8 -> float (4), 1 - a, 1 - b (2),
# eight products (8), three sums (3), cvRound (1)) = 24, + resize(INTER_AREA)'s multiply-add per window pixel (2) = 26 lane-ops.
# valu_insts_issued / (samples x 26 / 64) says how far the kernels are from that floor.
DESC_OPS_LOWER_BOUND = 26
MIN_WARM_S = float(os.environ.get("VFSMS_BENCH_MIN_WARM", "1.5"))     # 0 under rocprofv3 --pmc (serialised kernels make every step slow)


def pmc_value(kernel, key):
    """A per-launch PMC figure of `kernel` from the newest committed summary (profiles/*_pmc_summary.txt, written by
    tools/profile_round.sh from separate rocprofv3 --pmc passes of this same command).  PMC counters cannot be collected
    from inside the timed run, hence the file."""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.txt")), reverse=True):
        for line in open(path):
            if line.startswith(kernel + " ") and ("launches=" in line or kernel == "pmc_run"):
                m = re.search(re.escape(key) + r"=([0-9.e+]+)", line)
                if m:
                    return float(m.group(1)), os.path.relpath(path, ROOT)
    return None, None


def pmc_build():
    """(what the newest PMC summary was taken of, what this run loaded, stale?)  tools/profile_round.sh stamps the summary with the content
    hash of the kernel sources, the hash of the library and the commit (tools/build_id.py); the same hashes of THIS tree are formed here.
    stale = the kernel sources changed behind the profile (or the summary predates the stamp): the PMC-derived fields of the line
    (traffic, valu_busy_frac_pmc, ta_busy_frac_pmc, valu_issued_over_lower_bound, effective_clock_ghz) then describe an OLDER build and
    the line says so -- `pmc_stale: true` -- instead of passing them off as this run's."""
    import glob
    from tools.build_id import build_id
    mine = build_id(os.environ.get("VFSMS_LIB") or None)
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.txt")), reverse=True)
    theirs, src = {}, None
    if paths:
        src = os.path.relpath(paths[0], ROOT)
        for line in open(paths[0]):
            if line.startswith("# build:"):
                theirs = dict(tok.split("=", 1) for tok in line[len("# build:"):].split() if "=" in tok)
                break
    stale = theirs.get("src_sha256") != mine["src_sha256"]
    return dict(pmc_summary=src, pmc_build=theirs or None, this_build=mine, pmc_stale=bool(stale))


def pmc_total(kernel, key):
    """per-launch figure x launches of `kernel` in the newest PMC summary -> (total, launches)"""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.txt")), reverse=True):
        for line in open(path):
            if line.startswith(kernel + " ") and "launches=" in line:
                m = re.search(re.escape(key) + r"=([0-9.e+]+)", line)
                n = re.search(r"launches=(\d+)", line)
                if m and n:
                    return float(m.group(1)) * int(n.group(1)), int(n.group(1))
    return None, None


def pmc_traffic(kernel):
    return pmc_value(kernel, "total(x2 rule)")


def pmc_traffic_scaled(kernel, attempts_per_group):
    """HBM traffic of `kernel` for ONE launch group of this run: the counter total of the PMC run / the attempts that run evaluated (pmc_run
    line) x this run's attempts per group.  (Per-launch figures of the summary are per KERNEL launch of the PMC run: since round 4 a group
    launches an image-sized kernel once per shape run, and PMC runs need not have this run's batch sizes.)"""
    tot, _n = pmc_total(kernel, "total(x2 rule)")
    att, src = pmc_value("pmc_run", "attempts")
    if not (tot and att):
        return pmc_traffic(kernel)
    return tot / (att + 0.5) * attempts_per_group, src


def valu_over_bound(spk_live):
    """SQ_INSTS_VALU of k_describe + k_describe_small over the whole PMC run / (the keypoints that run described x samples per keypoint x 26
    lane-ops / 64 lanes): counter totals against an algorithmic count, independent of launch sizes"""
    a, _n = pmc_total("k_describe", "INSTS_VALU")
    b, _n = pmc_total("k_describe_small", "INSTS_VALU")
    kp, _s = pmc_value("pmc_run", "keypoints")
    sp, _s = pmc_value("pmc_run", "samples_per_keypoint")
    if not (a and kp):
        return None
    return round((a + (b or 0.0)) / (kp * (sp or spk_live) * DESC_OPS_LOWER_BOUND / 64.0), 3)


def clock_fields(achieved_tlaneops):
    """The VALU roof at the clock the kernel RUNS at: SQ_BUSY_CYCLES / 32 over the traced duration of k_describe in the PMC pass
    (eff_clock_ghz in the summary; ~2.0-2.1 GHz under this load against the 2.4 GHz of the peak constant)."""
    eff, _src = pmc_value("k_describe", "eff_clock_ghz")
    if not eff:
        return dict(effective_clock_ghz=None, peak_at_effective_clock=None, frac_at_effective_clock=None)
    peak = 256 * 4 * 16 * eff * 1e9 / 1e12
    return dict(effective_clock_ghz=round(eff, 3), peak_at_effective_clock=round(peak, 2), frac_at_effective_clock=round(achieved_tlaneops / peak, 4),
                peak_clock_note="peak / frac assume 2.4 GHz; *_at_effective_clock use the shader clock measured under this kernel in the PMC pass")


def sum_or_none(vals):
    return sum(vals) if all(v is not None for v in vals) else None


def hbm_roofline(kernel, bytes_per_launch, ms_per_launch, launches, note=None, **extra):
    dur = ms_per_launch * 1e-3
    gbs = bytes_per_launch / dur / 1e9 if dur > 0 else 0.0
    traffic, src = pmc_traffic(kernel.split("+")[0].split(" ")[0])
    d = dict(kernel=kernel, bound="hbm", achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 5),
             traffic=traffic, traffic_source=src, bytes_per_launch=bytes_per_launch, avg_launch_ms=round(ms_per_launch, 4), launches=launches)
    if note:
        d["note"] = note
    d.update(extra)
    return d


def optimal_dft_size(n):
    m = max(int(n), 1)
    while True:
        k = m
        for p in (2, 3, 5):
            while k % p == 0:
                k //= p
        if k == 1:
            return m
        m += 1


def bench_fuse(args, eng, grid, tiles, handles, torch, close=True):
    """Secondary metric: mosaic assembly of the whole grid from its true offsets -- layout arithmetic of
    Stitcher.getStitchByOffset, tile 0 pasted, every further tile blended into the device canvas with fadeInAndFadeOut
    (strip mode in columns, corner mode after each serpentine turn).  `value`: tiles already resident in HBM (the handles the
    registration phase uploaded; what Stitcher.flowStitch does for gray mosaics); the host-tile variant of the same calls (one
    H2D copy per tile) and the final canvas download are timed beside it."""
    import imagestitch_amd as isa
    if args.gpus != 1:
        raise SystemExit("--method fuse is a single-GPU measurement (the canvas is order-dependent: replicas only)")
    n = grid.n_tiles
    offs = [[0, 0]] + [list(map(int, o)) for o in grid.true_offsets()]
    shapes = [(grid.th, grid.tw)] * n
    offsetList, rangeX, rangeY, rows, cols = isa.Stitcher._layout(shapes, offs)
    rois = []
    for i in range(1, n):
        oy, ox = offsetList[i]
        rois.append((max(oy, rangeX[i - 1][0]), max(ox, rangeY[i - 1][0]), min(oy + grid.th, rangeX[i - 1][1]), min(ox + grid.tw, rangeY[i - 1][1])))

    big = n > 128          # configs[4]: a 118 k x 118 k px mosaic (13.9 GB): no whole-canvas download, no second pass from host tiles
    dl_s = [0.0]           # seconds of the last assemble()'s download

    def assemble(download, resident=True):
        canvas = eng.canvas_create(rows, cols, 1)
        try:
            if resident:                                   # what Stitcher.getStitchByOffset calls for resident tiles: the whole walk, one call
                geom = [(offsetList[0][0], offsetList[0][1], 0, 0, 0, 0, 0, 0, -1)]
                geom += [(offsetList[i][0], offsetList[i][1]) + tuple(rois[i - 1]) + (offs[i][0], offs[i][1], 0) for i in range(1, n)]
                eng.canvas_assemble_resident(canvas, handles, geom)
            for i in range(0 if not resident else n, n):
                oy, ox = offsetList[i]
                if i == 0:
                    eng.canvas_paste_tile(canvas, handles[i], oy, ox) if resident else eng.canvas_paste(canvas, tiles[i], oy, ox)
                    continue
                if resident:
                    eng.canvas_fuse_tile_resident(canvas, handles[i], oy, ox, rois[i - 1], offs[i][0], offs[i][1])
                else:
                    eng.canvas_fuse_tile(canvas, tiles[i], oy, ox, rois[i - 1], offs[i][0], offs[i][1])
            eng.sync()
            t_dl = time.perf_counter()
            try:
                if download and big:                      # a band across the first serpentine turn stands for the mosaic
                    return eng.canvas_download_rows(canvas, 0, min(rows, grid.th), cols, 1)
                return eng.canvas_download(canvas, rows, cols, 1) if download else None
            finally:
                dl_s[0] = time.perf_counter() - t_dl
        finally:
            eng.canvas_free(canvas)

    for _ in range(max(args.warmup, 1)):
        assemble(False)
    # The timed steps run WITHOUT the engine's per-stage HIP events: a mosaic is ~180 dependent launches of a few microseconds each, and an
    # event record in front of and behind every tile's pair of launches (what vfsms_profile_* does) is a marker packet the command processor
    # has to retire before the next dispatch -- it was part of what rounds 2-5 reported as the fuse's "launch latency".  The stage times of
    # the roofline come from a second set of K steps with the events on (ms_per_step_with_stage_events).
    torch.cuda.synchronize(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        assemble(False)
    torch.cuda.synchronize(); eng.sync()
    dt = (time.perf_counter() - t0) / args.steps
    eng.profile_enable(True); eng.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        assemble(False)
    torch.cuda.synchronize(); eng.sync()
    dt_prof = (time.perf_counter() - t0) / args.steps
    prof = eng.profile_read(reset=True)
    eng.profile_enable(False)
    out = assemble(True)
    dl = dl_s[0]
    dt_host = None
    if not big:
        t2 = time.perf_counter()
        out_host = assemble(True, resident=False)
        dt_host = time.perf_counter() - t2 - dl_s[0]
        assert np.array_equal(out, out_host)
    mpx = rows * cols / 1e6
    # algorithmic bytes: per fused tile read canvas ROI + read tile ROI + write ROI (3 r c) plus the paste of the
    # tile's pixels outside the ROI (read + write); the validity plane adds r c / 8 -- not counted
    fuse_bytes = sum(3 * (r[2] - r[0]) * (r[3] - r[1]) + 2 * (grid.th * grid.tw - (r[2] - r[0]) * (r[3] - r[1])) for r in rois) + 2 * grid.th * grid.tw
    f_ms, f_n = prof.get("fuse", (0.0, 0))
    roof = None
    if f_n:
        roof = hbm_roofline("k_fuse_counts_pick+k_fuse_apply", fuse_bytes, min(f_ms / args.steps, dt * 1e3), args.steps,
                            note="one 'launch' = the whole mosaic (%d tiles; per tile: the count of non-zero elements per quadrant + the pick of "
                                 "getWeightsMatrix's geometry, then the blend -- a strip tile the blend alone; the validity of a ROI comes from the "
                                 "canvas's rectangle list on the host); bytes = 3 r c per fuse ROI + 2 B/px pasted outside it; the walk is a chain of "
                                 "dependent ~12 us launches, not a stream" % n, launch_groups=f_n)
        # PMC traffic of one mosaic: per-launch traffic x launches per mosaic of the walk's kernels (profiles/*_pmc_summary.txt, mosaic section)
        tot, src = 0.0, None
        for kname in ("k_fuse_apply", "k_fuse_counts_pick<16, 16>", "k_paste"):
            t_k, n_k = pmc_total(kname, "total(x2 rule)")
            m_k, src_k = pmc_value(kname, "mosaics")
            if t_k is None or not m_k:
                tot = None
                break
            tot += t_k / m_k
            src = src_k
        if tot:
            roof["traffic"] = tot
            roof["traffic_source"] = src
            roof["traffic_over_algorithmic"] = round(tot / fuse_bytes, 3)
    print(json.dumps({
        "metric": "fuse Mpx/sec (mosaic pixels, fadeInAndFadeOut)", "value": round(mpx / dt, 2), "unit": "Mpx/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "ms_per_step_with_stage_events": round(dt_prof * 1e3, 3),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "synthetic %dx%d grid of %dx%d u8 tiles assembled from the ground-truth offsets, fadeInAndFadeOut, mosaic %dx%d"
                               % (args.rows, args.cols, args.tile, args.tile, rows, cols), "tiles": n, "tiles_resident_in_hbm": True,
                   "canvas_download_ms": round(dl * 1e3, 1), "canvas_download_is": "the first %d rows" % min(rows, grid.th) if big else "the whole mosaic",
                   "ms_per_step_with_host_tiles": round(dt_host * 1e3, 1) if dt_host is not None else None},
        "roofline": roof, "cpu_baseline": (cpu_baseline_fuse(args, grid, tiles, offsetList, rois, offs, rows, cols) if args.cpu_sample > 0 else None),
        "stages": {k: dict(ms=round(v[0], 3), launches=v[1]) for k, v in prof.items()},
        "tile_Mpx_per_s": round(n * grid.th * grid.tw / 1e6 / dt, 2), "mosaic_nonzero_fraction": round(float((out > 0).mean()), 4),
        "mosaic_Mpx_per_s_of_the_fuse_kernels": round(mpx / (f_ms / args.steps * 1e-3), 2) if f_n else None,
        "note": "value = mosaic pixels / wall clock of one assembly INCLUDING the canvas allocation + clear and its release (at configs[4]'s 13.9 GB canvas "
                "that is most of the step: the fuse kernels themselves take stages.fuse)"}))
    if close:
        eng.close()


def _jsonline(d):
    print(json.dumps(d))


def bench_dendritic25(args, eng, torch):
    """Second SURF line: the 25 committed pairs of the reference's dendriticCrystal set (five 6-tile neighbourhoods around the serpentine
    turns, tests/golden/real_path_strips.*: 1936 x 2584 frames rebuilt around the 640-px crops of the ROI strips the accepted attempts
    read).  Real dendrite texture carries ~1.6x the SURF keypoints per pixel of the synthetic grid, so BF work per attempt is what
    configs[1] really costs.  One step = all five neighbourhoods through GridRegistrar (direction threaded inside each)."""
    import imagestitch_amd as isa
    from imagestitch_amd.grid import GridRegistrar
    if args.gpus != 1 or args.method != "surf":
        raise SystemExit("--workload dendritic25 is a single-GPU SURF measurement")
    gd = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gd, "real_path_strips.json")))["neighbourhoods"]
    z = np.load(os.path.join(gd, "real_path_strips.npz"))
    nbs = []
    for nb in meta:
        H, W = nb["shape"]
        frames = {t: np.zeros((H, W), np.uint8) for t in nb["tiles"]}
        for s_ in nb["strips"]:
            a = z[s_["key"]]
            frames[s_["tile"]][s_["y0"]:s_["y0"] + a.shape[0], s_["x0"]:s_["x0"] + a.shape[1]] = a
        fr = [frames[t] for t in nb["tiles"]]
        nbs.append(dict(nb=nb, hs=[eng.tile_upload(f) for f in fr], shapes=[f.shape for f in fr]))
    # one registrar per neighbourhood: each remembers ITS path (the five have the same length and different turns -- a shared memory would hand
    # every neighbourhood the previous one's pattern, which costs attempts)
    regs = [GridRegistrar(eng, method="surf", roiRatio=0.2, searchRatio=0.75, offsetEvaluate=3, directIncre=1, surfParams=eng.surf_params(), window=args.window)
            for _ in nbs]

    reg = regs[0]

    def step():
        return [r.register(n["hs"], n["shapes"], n["nb"]["incoming_direction"])[0] for r, n in zip(regs, nbs)]
    tables = step()
    worst = 0
    for n, tb in zip(nbs, tables):
        for row, e in zip(tb, n["nb"]["expected"]):
            assert row[0] == 1 and [int(row[1]), int(row[2])] == e["offset"], (e, row)
            worst = max(worst, abs(int(row[1]) - e["gold"][0]), abs(int(row[2]) - e["gold"][1]))
    t_w = time.perf_counter()
    warm_extra = 0
    for _ in range(args.warmup):
        step()
    while time.perf_counter() - t_w < MIN_WARM_S and warm_extra < 400:
        step(); warm_extra += 1
    for r in regs:
        for k in r.stats:
            r.stats[k] = 0
    eng.profile_enable(True); eng.profile_read(reset=True)
    torch.cuda.synchronize(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(); eng.sync()
    dt = (time.perf_counter() - t0) / args.steps
    prof = eng.profile_read(reset=True); eng.profile_enable(False)
    st = {k: sum(r.stats[k] for r in regs) for k in reg.stats}
    stages = {k: dict(ms=round(v[0], 3), launches=v[1], ms_per_launch=round(v[0] / max(v[1], 1), 4)) for k, v in prof.items()}
    P = 25
    bf_ms, bf_n = prof.get("bf_mfma", (0.0, 0))
    de_ms, de_n = prof.get("describe", (0.0, 0))
    # A FIXTURE-SHAPED line, not configs[1]'s cost: 640-px crops of the accepted ROI strips in otherwise empty frames give 2.6 k keypoints per ROI
    # and many small batches.  No roofline is claimed for it; what it establishes is the keypoint DENSITY of real dendrite texture (per kpx of
    # textured ROI), which equals the synthetic grid's -- the headline workload is representative per pixel.
    roof = None
    _jsonline({"metric": "image-pairs/sec (FIXTURE: 640-px crops of dendriticCrystal ROI strips in empty frames, SURF+BF; not configs[1]'s cost)", "value": round(P / dt, 3), "unit": "image-pairs/s", "n_gpus": 1,
               "steps": args.steps, "warmup": args.warmup, "warmup_extra_steps_until_1p5s": warm_extra, "ms_per_step": round(dt * 1e3, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "real (committed crops of the reference's demo tiles)",
               "config": {"workload": "the 25 committed pairs of demoImages/dendriticCrystal (5 neighbourhoods x 6 tiles of 1936x2584, 640-px crops of the "
                                      "accepted ROI strips in zero frames); SURF(100,4,3,64-d)+BF-L2 knn2 ratio 0.75 + mode vote; roiRatio 0.2, offsetEvaluate 3, directIncre 1",
                          "pairs": P, "parallelism": "pairs1"},
               "max_abs_offset_error_px_vs_stitcher_py_87": worst, "rows_equal_oracle": True,
               "attempts_per_step": st["attempts"] / max(args.steps, 1), "batches_per_step": st["batches"] / max(args.steps, 1),
               "keypoints_per_roi": round(st["sum_nq_plus_nt"] / max(2 * st["attempts"], 1), 1),
               "keypoints_per_kpx_of_textured_roi": round(st["sum_nq_plus_nt"] / max(2 * st["attempts"], 1) / (640 * 387 / 1000.0), 2),
               "note": "fixture-shaped: rows == oracle and within 1 px of Stitcher.py:87 on real texture; keypoints_per_kpx_of_textured_roi is the figure to "
                       "compare with the synthetic grid (10.4 per kpx) -- the rate and the stage times describe the fixture, not the dataset",
               "roofline": roof, "cpu_baseline": None, "stages": stages})
    eng.close()


def bench_line_scan(args, eng, torch):
    """The path 4 of the reference's 6 demo datasets take (Main.py:29-51): calculateOffsetForFeatureSearch over a line scan of 24 tiles of
    1024 x 1280 (zirconCL's geometry; synthetic texture with a burned-in data bar): whole-tile SURF, B -> A feature reuse.  Two numbers:
    the pair-by-pair loop through the reference's call surface (one launch sequence + two host synchronisations per tile) and the
    batched form flowStitch uses (16 tiles per fused launch sequence, all matches in one batch)."""
    import imagestitch_amd as isa
    from imagestitch_amd.synthetic import line_scan
    if args.gpus != 1:
        raise SystemExit("--method surf_full is a single-GPU measurement")
    tiles, truth = line_scan(n=24)
    P = len(tiles) - 1
    old = (isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.featureMethod, isa.Stitcher.offsetEvaluate, isa.Stitcher.isEnhance)
    isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.featureMethod, isa.Stitcher.offsetEvaluate, isa.Stitcher.isEnhance = 4, 0, "surf", 3, False
    try:
        s = isa.Stitcher(); s._engine = eng; s.isPrintLog = False
        hs = [eng.tile_upload(t) for t in tiles]

        def loop():
            s.tempImageFeature.isBreak = True
            out = [s.calculateOffsetForFeatureSearch([tiles[k], tiles[k + 1]]) for k in range(P)]
            s.releaseTiles()
            return out

        def batched():
            return s._fullImageTable(hs)
        want = loop(); got = batched()
        assert all(w[0] for w in want) and [[r[1], r[2]] for r in got] == [w[1] for w in want]
        worst = max(max(abs(r[1] - t_[0]), abs(r[2] - t_[1])) for r, t_ in zip(got, truth))
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < MIN_WARM_S:
            batched()
        eng.profile_enable(True); eng.profile_read(reset=True)
        torch.cuda.synchronize(); eng.sync(); t0 = time.perf_counter()
        for _ in range(args.steps):
            batched()
        torch.cuda.synchronize(); eng.sync()
        dt = (time.perf_counter() - t0) / args.steps
        prof = eng.profile_read(reset=True); eng.profile_enable(False)
        loop(); eng.sync(); t0 = time.perf_counter()
        for _ in range(args.steps):
            loop()
        eng.sync()
        dt_loop = (time.perf_counter() - t0) / args.steps
    finally:
        isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.featureMethod, isa.Stitcher.offsetEvaluate, isa.Stitcher.isEnhance = old
    stages = {k: dict(ms=round(v[0], 3), launches=v[1], ms_per_launch=round(v[0] / max(v[1], 1), 4)) for k, v in prof.items()}
    _jsonline({"metric": "image-pairs/sec (1024x1280 line scan, whole-tile SURF+BF, feature reuse)", "value": round(P / dt, 3), "unit": "image-pairs/s", "n_gpus": 1,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "24-tile line scan of 1024x1280 u8 tiles (20 % overlap, burned-in data bar), calculateOffsetForFeatureSearch: whole-tile "
                                      "SURF(100,4,3,64-d), every tile described once, BF-L2 knn2 ratio 0.75 + mode vote; batched: 16 tiles per launch sequence, 23 matches in one batch",
                          "pairs": P, "parallelism": "pairs1"},
               "max_abs_offset_error_px": int(worst),
               "pair_by_pair_loop_pairs_per_s": round(P / dt_loop, 3), "pair_by_pair_loop_ms_per_step": round(dt_loop * 1e3, 3),
               "batched_over_loop": round(dt_loop / dt, 2), "roofline": None, "cpu_baseline": None, "stages": stages})
    eng.close()


def bench_from_files(args, eng, grid, torch):
    """Decode-inclusive registration (scope row f-1): the grid's tiles as JPEG files on disk, registered through the Stitcher's own
    entry (Stitcher._registerBatched: device handles reserved up front, a pool of decoder threads fills them while the native registrar
    already works on the first pairs).  Reported beside the decode-only rate of the same pool and the resident-tiles registration rate:
    the pipeline is good when end-to-end is close to the slower of the two."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    import imagestitch_amd as isa
    from imagestitch_amd import stitcher as ST
    if args.gpus != 1 or args.method != "surf":
        raise SystemExit("--from-files is a single-GPU SURF measurement")
    # (the Stitcher's own default: 32 decoder threads with the library's JPEG decoder, 16 with Pillow -- Stitcher._decoderThreads)
    nthreads = max(1, min(args.decode_threads or min(os.cpu_count() or 4, 32 if os.environ.get("VFSMS_NATIVE_JPEG", "1") != "0" else 16), grid.n_tiles, 64))
    color = bool(args.color)
    with tempfile.TemporaryDirectory(prefix="vfsms_bench_") as d:
        files = []
        for k, t in enumerate(grid.tiles(range(grid.n_tiles), threads=min(8, os.cpu_count() or 1))):
            f = os.path.join(d, "tile_%03d.jpg" % k)
            if color:                                         # a tinted colour version: Cb / Cr planes that are not flat (4:2:0 like camera JPEGs)
                ft = t.astype(np.float32)
                t = np.clip(np.stack([0.6 * ft + 30, ft, 255 - 0.7 * ft], -1), 0, 255).astype(np.uint8)
            Image.fromarray(t).save(f, quality=90)
            files.append(f)
        mb = sum(os.path.getsize(f) for f in files) / 1e6
        s = isa.Stitcher(); s._engine = eng; s.isPrintLog = False
        old = (isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.featureMethod, isa.Stitcher.offsetEvaluate,
               isa.Stitcher.isColorMode, isa.Stitcher.fuseMethod)
        isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.featureMethod, isa.Stitcher.offsetEvaluate = 1, 0.2, "surf", args.offset_evaluate
        # --color: Main.py:14's default -- ONE decode per file gives the registration plane and the mosaic's B G R tile (vfsms_tile_fill_pair);
        # the colour tiles stay resident for getStitchByOffset and are released here after every step.  Without it: gray tiles only.
        isa.Stitcher.isColorMode, isa.Stitcher.fuseMethod = color, "notFuse"
        s.decodeThreads = nthreads
        truth = grid.true_offsets()

        def e2e():
            s.direction = 1
            out = s._registerBatched(files, s.calculateOffsetForFeatureSearchIncre)
            for h, _shape in (s.__dict__.pop("_resident", None) or {}).values():
                eng.tile_free(h)
            return out
        try:
            status, end, offs, _desc = e2e()
            assert status and end == grid.n_pairs, (status, end)
            worst = max(max(abs(o[0] - t_[0]), abs(o[1] - t_[1])) for o, t_ in zip(offs, truth))
            for _ in range(max(args.warmup, 1)):
                e2e()
            torch.cuda.synchronize(); eng.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                e2e()
            torch.cuda.synchronize(); eng.sync()
            dt = (time.perf_counter() - t0) / args.steps
            ist = dict(getattr(s, "_ingestStats", {}) or {})
            # decode only, same pool size, the decoder the pipeline used: the library's own (libjpeg-turbo into a buffer kept per thread) when
            # every tile of the last step went through vfsms_tile_fill_jpeg, else Pillow (with the block allocator the Stitcher's pool sets)
            native = ist.get("native", 0) == ist.get("tiles", -1)
            if native:
                import ctypes, threading
                tl = threading.local()

                def decode_only(f):
                    if not hasattr(tl, "buf"):
                        tl.buf = np.empty(grid.th * grid.tw * 3, np.uint8)
                    data = open(f, "rb").read()
                    h_, w_, c_ = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                    rc = eng.lib.vfsms_jpeg_decode(data, len(data), int(color), tl.buf.ctypes.data_as(ctypes.c_void_p), tl.buf.nbytes,
                                                   ctypes.byref(h_), ctypes.byref(w_), ctypes.byref(c_))
                    assert rc == 0 and (h_.value, w_.value) == (grid.th, grid.tw), rc
            else:
                def decode_only(f):
                    ST._decode_once(f, color)
            with ST._PillowBlocks(color and not native), ThreadPoolExecutor(max_workers=nthreads) as ex:
                list(ex.map(decode_only, files[:nthreads]))
                t0 = time.perf_counter()
                list(ex.map(decode_only, files))
                dt_dec = time.perf_counter() - t0
            t0 = time.perf_counter()
            decode_only(files[0])
            dt_one = time.perf_counter() - t0
            # registration only (tiles resident)
            from imagestitch_amd.grid import GridRegistrar
            hs = [eng.tile_upload(ST._imread(f, False)) for f in files]
            reg = GridRegistrar(eng, method="surf", roiRatio=0.2, searchRatio=0.75, offsetEvaluate=args.offset_evaluate, directIncre=1,
                                surfParams=eng.surf_params(), window=48)
            shapes = [(grid.th, grid.tw)] * grid.n_tiles
            reg.register(hs, shapes, 1)
            reg.register(hs, shapes, 1)                       # (the second one runs on the learned scan pattern: arena and clocks settled)
            eng.sync(); t0 = time.perf_counter()
            for _ in range(args.steps):
                reg.register(hs, shapes, 1)
            eng.sync()
            dt_reg = (time.perf_counter() - t0) / args.steps
            for h in hs:
                eng.tile_free(h)
        finally:
            (isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.featureMethod, isa.Stitcher.offsetEvaluate,
             isa.Stitcher.isColorMode, isa.Stitcher.fuseMethod) = old
    P = grid.n_pairs
    slower = max(dt_dec, dt_reg)
    _jsonline({"metric": "image-pairs/sec, decode inclusive (2048x2048 %s JPEG files, SURF+BF)" % ("colour" if color else "grayscale"), "value": round(P / dt, 3), "unit": "image-pairs/s",
               "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic (JPEG quality 90 on local disk, %.0f MB for %d tiles)" % (mb, grid.n_tiles),
               "config": {"workload": "synthetic %dx%d grid of %dx%d tiles as %s JPEG files -> Stitcher ingest pipeline (vfsms_tile_reserve / %s, "
                                      "%d decoder threads, one decode per file) -> native registrar; SURF+BF-L2+mode as the default workload"
                                      % (args.rows, args.cols, args.tile, args.tile, "colour (isColorMode = True: gray plane + resident B G R tile from the same decode)" if color else "grayscale",
                                         "vfsms_tile_fill_jpeg" if native else "vfsms_tile_fill_pair" if color else "vfsms_tile_fill", nthreads),
                          "pairs": P, "decode_threads": nthreads, "host_cores": os.cpu_count(), "color": color,
                          "decoder": ("libjpeg-turbo inside libvfsms (vfsms_tile_fill_jpeg: pinned staging, no interpreter lock)" if native
                                      else "Pillow (VFSMS_NATIVE_JPEG=0 or no libjpeg.so.8) + %s" % ("vfsms_tile_fill_pair" if color else "vfsms_tile_fill"))},
               "max_abs_offset_error_px": int(worst),
               "decode_only_ms_per_step": round(dt_dec * 1e3, 2), "decode_only_tiles_per_s": round(grid.n_tiles / dt_dec, 1),
               "decode_one_tile_one_thread_ms": round(dt_one * 1e3, 2),
               "registration_only_ms_per_step": round(dt_reg * 1e3, 2), "registration_only_pairs_per_s": round(P / dt_reg, 1),
               "end_to_end_over_slower_stage": round(dt / slower, 3),
               "ingest_thread_ms_per_tile": (dict(decode=round(ist["decode_s"] / max(ist["tiles"], 1) * 1e3, 2), hand_over=round(ist["fill_s"] / max(ist["tiles"], 1) * 1e3, 2))
                                             if ist.get("tiles") else None),
               "startup_bubble_note": "a path cannot start before its first tiles are decoded: end-to-end >= one tile's decode latency + the registration time",
               "roofline": None, "cpu_baseline": None})
    eng.close()


def cpu_baseline_surf(args, grid, tiles, isa):
    """The oracle (a port: cv2 is not installable) on a bounded sample of the same grid, built -O3 -march=native ON this box
    (oracle/Makefile `native`): single-thread row and all-host-threads row.  One ROI attempt per pair at the true
    direction; reported, never credited."""
    from oracle import oracle as O
    O.build()
    try:
        O.use_native()
    except Exception as e:                      # no compiler on the box: the portable build is timed instead
        print("cpu_baseline: native oracle build unavailable (%s)" % e, file=sys.stderr)
    cores = os.cpu_count() or 1
    dirs = grid.true_directions()
    S = min(args.cpu_sample, grid.n_pairs)

    def run(pairs, nthreads):
        t1 = time.perf_counter()
        for k in pairs:
            A, B = tiles[k], tiles[k + 1]
            ra = isa.roi_rect(A.shape, dirs[k], "first", 0.2); rb = isa.roi_rect(B.shape, dirs[k], "second", 0.2)
            ka, da = O.surf_detect_describe(np.ascontiguousarray(A[ra[0]:ra[0] + ra[2], ra[1]:ra[1] + ra[3]]), nthreads=nthreads)
            kb, db = O.surf_detect_describe(np.ascontiguousarray(B[rb[0]:rb[0] + rb[2], rb[1]:rb[1] + rb[3]]), nthreads=nthreads)
            pr = O.bf_l2_ratio_matches(da, db, 0.75, nthreads=nthreads)
            O.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pr, 3)
        return time.perf_counter() - t1
    # all-cores row: ONE PAIR PER THREAD (the pairs are independent -- what a multi-process CPU deployment of the reference would do);
    # OpenMP inside a pair barely scales (1.9x on 256 threads), so that is not the fair all-cores number
    from concurrent.futures import ThreadPoolExecutor
    n_par = min(grid.n_pairs, max(cores, 1))
    pair_ids = [k % len(tiles) for k in range(n_par)] if not isinstance(tiles, dict) else [k for k in sorted(tiles) if k + 1 in tiles][:n_par]
    t1 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(cores, len(pair_ids))) as ex:
        list(ex.map(lambda k: run([k], 1), pair_ids))
    dt_all = time.perf_counter() - t1
    S1 = max(1, min(S, 6))
    dt_one = run(pair_ids[:S1], 1)
    return dict(value=round(len(pair_ids) / dt_all, 4), unit="image-pairs/s", cores=min(cores, len(pair_ids)), kind="port",
                sample="%d pairs of the same grid, one pair per host thread (%d threads of %d), one ROI attempt each at the true direction (oracle "
                       "SURF+BF-L2+mode built -O3 -march=native here), %.1f s" % (len(pair_ids), min(cores, len(pair_ids)), cores, dt_all),
                single_thread=dict(value=round(S1 / dt_one, 4), cores=1, sample="first %d pair(s), %.1f s" % (S1, dt_one)),
                build=O.build_kind())


def cpu_baseline_pairs(args, grid, tiles, isa, method):
    """cpu_baseline of the ORB / phase lines: the oracle's chain for ONE ROI attempt per pair at the true direction (ImageUtility.py:260-262 +
    297-302 + 139-178 for orb, Stitcher.py:230 for phase), one pair per host thread and a single-thread row, as cpu_baseline_surf does."""
    from oracle import oracle as O
    O.build()
    try:
        O.use_native()
    except Exception as e:
        print("cpu_baseline: native oracle build unavailable (%s)" % e, file=sys.stderr)
    cores = os.cpu_count() or 1
    dirs = grid.true_directions()

    def one(k):
        A, B = tiles[k], tiles[k + 1]
        ra = isa.roi_rect(A.shape, dirs[k], "first", 0.2); rb = isa.roi_rect(B.shape, dirs[k], "second", 0.2)
        roiA = np.ascontiguousarray(A[ra[0]:ra[0] + ra[2], ra[1]:ra[1] + ra[3]]); roiB = np.ascontiguousarray(B[rb[0]:rb[0] + rb[2], rb[1]:rb[1] + rb[3]])
        if method == "phase":
            O.phase_correlate(roiA, roiB)
            return
        ka, da = O.orb_detect_describe(roiA); kb, db = O.orb_detect_describe(roiB)
        pr, _dist = O.bf_hamming_matches(da, db)
        O.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pr, args.offset_evaluate)
    from concurrent.futures import ThreadPoolExecutor
    pair_ids = [k for k in sorted(tiles) if k + 1 in tiles][:min(grid.n_pairs, cores)]
    t1 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(cores, len(pair_ids))) as ex:
        list(ex.map(one, pair_ids))
    dt_all = time.perf_counter() - t1
    S1 = max(1, min(args.cpu_sample, 6, len(pair_ids)))
    t1 = time.perf_counter()
    for k in pair_ids[:S1]:
        one(k)
    dt_one = time.perf_counter() - t1
    what = "oracle cv2.phaseCorrelate restatement (FP64, own mixed-radix FFT)" if method == "phase" else "oracle ORB(5000,1.2,8)+BF-Hamming 1-NN+mode"
    return dict(value=round(len(pair_ids) / dt_all, 4), unit="image-pairs/s", cores=min(cores, len(pair_ids)), kind="port",
                sample="%d pairs of the same grid, one pair per host thread (%d threads of %d), one ROI attempt each at the true direction (%s, built "
                       "-O3 -march=native here), %.1f s" % (len(pair_ids), min(cores, len(pair_ids)), cores, what, dt_all),
                single_thread=dict(value=round(S1 / dt_one, 4), cores=1, sample="first %d pair(s), %.1f s" % (S1, dt_one)), build=O.build_kind())


def cpu_baseline_fuse(args, grid, tiles, offsetList, rois, offs, rows, cols):
    """cpu_baseline of the fuse line: the reference's walk (Stitcher.py:434-486: int64 canvas with -1 for empty, every tile pasted, its
    overlap with the canvas blended by ImageFusion.fuseByFadeInAndFadeOut -- the oracle's C restatement) over the FIRST tiles of the same
    mosaic, bounded to ~20 s on one host thread (the walk is a chain: tile i blends against what tiles 0..i-1 left).  Mpx/s of the mosaic
    area those tiles cover."""
    from oracle import oracle as O
    O.build()
    try:
        O.use_native()
    except Exception as e:
        print("cpu_baseline: native oracle build unavailable (%s)" % e, file=sys.stderr)
    n = min(grid.n_tiles, max(2, int(os.environ.get("VFSMS_BENCH_FUSE_CPU_TILES", "128"))))     # (the 25-s bound below cuts larger mosaics)
    r1 = max(offsetList[i][0] + grid.th for i in range(n)); c1 = max(offsetList[i][1] + grid.tw for i in range(n))
    r0 = min(offsetList[i][0] for i in range(n)); c0 = min(offsetList[i][1] for i in range(n))
    t1 = time.perf_counter()
    canvas = np.full((r1 - r0, c1 - c0), -1, np.int64)
    done = 0
    for i in range(n):
        oy, ox = offsetList[i][0] - r0, offsetList[i][1] - c0
        T = np.asarray(tiles[i]).astype(np.int64)
        if i > 0:
            y0, x0, y1, x1 = rois[i - 1]
            y0 -= r0; y1 -= r0; x0 -= c0; x1 -= c0
            A = canvas[y0:y1, x0:x1].copy()
            B = T[y0 - oy:y1 - oy, x0 - ox:x1 - ox]
            fused = O.fuse_fade(A, B, offs[i][0], offs[i][1])
            canvas[oy:oy + grid.th, ox:ox + grid.tw] = T
            canvas[y0:y1, x0:x1] = fused
        else:
            canvas[oy:oy + grid.th, ox:ox + grid.tw] = T
        done = i + 1
        if time.perf_counter() - t1 > 25.0:
            break
    dt = time.perf_counter() - t1
    area = float((canvas >= 0).sum()) / 1e6
    return dict(value=round(area / dt, 2), unit="Mpx/s", cores=1, kind="port",
                sample="the reference walk (int64 / -1 canvas, paste + oracle fuseByFadeInAndFadeOut per overlap) over the first %d of %d tiles of "
                       "the same mosaic: %.1f Mpx in %.1f s on one host thread (the walk is a dependency chain)" % (done, grid.n_tiles, area, dt),
                build=O.build_kind())


def project_shards(args, eng, reg, handles, shapes, res, P, ms1_s, fence, gather, progress):
    """--project-shards: the pair-sharded step of N ranks on ONE GPU, rank after rank (GridRegistrar.register_projected), K steps per N.
    What it shows: the work split, the launch chain, readback and interpreter time of a rank whose chunk is 1 / N of the path -- the fixed
    cost per rank that decides the efficiency at N = 8 -- on the real kernels.  What it cannot show: RCCL at N ranks (the gather is run at
    world size 1 when --force-dist is given, else left out and said so), xGMI, N host processes sharing the box's cores."""
    from imagestitch_amd.grid import GridRegistrar
    out = {}
    K = max(args.steps, 1)
    for N in [int(v) for v in args.project_shards.split(",") if v.strip()]:
        rp = GridRegistrar(eng, method=args.method, roiRatio=0.2, searchRatio=0.75, offsetEvaluate=args.offset_evaluate, directIncre=1,
                           surfParams=reg.params, window=args.window)
        rp.path_memory = list(reg.path_memory) if reg.path_memory is not None else None      # what the session has learned (as on every rank)
        rp._kp_cap = getattr(reg, "_kp_cap", 0)
        eng.profile_enable(True); eng.profile_read(reset=True)

        def probe():
            eng.sync()
            return sum(v[0] for v in eng.profile_read(reset=True).values())
        full, _d, pr, tail = rp.register_projected(handles, shapes, 1, N, probe, gather)      # warm (arena sizes of the small batches)
        assert np.array_equal(full, res), "the projected sharded form does not reproduce the one-GPU table"
        acc = [dict(wall=0.0, gpu=0.0, attempts=0, batches=0, repair=0.0) for _ in range(N)]
        tails = 0.0
        fence()
        for _ in range(K):
            full, _d, pr, tail = rp.register_projected(handles, shapes, 1, N, probe, gather)
            tails += tail
            for r, q in enumerate(pr):
                acc[r]["wall"] += q["wall_s"]; acc[r]["gpu"] += q["probe"] or 0.0; acc[r]["attempts"] += q["attempts"]; acc[r]["batches"] += q["batches"]
                acc[r]["repair"] += q["repair_wall_s"]; acc[r]["pairs"] = q["pairs"]
        eng.profile_enable(False)
        ranks = [dict(rank=r, pairs=a["pairs"], attempts_per_step=a["attempts"] / K, batches_per_step=a["batches"] / K,
                      wall_ms=round(a["wall"] / K * 1e3, 3), gpu_ms=round(a["gpu"] / K, 3), host_ms=round((a["wall"] / K * 1e3) - a["gpu"] / K, 3),
                      repair_ms=round(a["repair"] / K * 1e3, 3)) for r, a in enumerate(acc)]
        slowest = max(q["wall_ms"] + q["repair_ms"] for q in ranks)
        tail_ms = tails / K * 1e3
        step_ms = slowest + tail_ms
        out["N=%d" % N] = dict(
            ranks=ranks, tail_ms_gather_assemble_learn=round(tail_ms, 3), slowest_rank_ms=round(slowest, 3), projected_ms_per_step=round(step_ms, 3),
            projected_pairs_per_s=round(P / (step_ms * 1e-3), 1), projected_efficiency_vs_this_run_at_1=round((ms1_s * 1e3) / (N * step_ms), 3),
            attempts_all_ranks=sum(q["attempts_per_step"] for q in ranks), repair_rounds=getattr(rp, "hint_repairs", 0),
            gather=("RCCL all_gather at world size 1 inside the tail (--force-dist)" if gather is not None else "not run (single process): add ~0.25 ms, "
                    "profiles/r05_bench_force_dist_rccl_world1.json"))
        progress("projected N=%d: slowest rank %.2f ms + tail %.2f ms -> %.0f pairs/s (efficiency %.2f)" % (N, slowest, tail_ms, P / (step_ms * 1e-3),
                                                                                                     (ms1_s * 1e3) / (N * step_ms)))
    out["note"] = ("PROJECTED from one GPU, not measured on N: every emulated rank runs its chunk alone on this device through the ranks' own code "
                   "(shard_payload / assemble / _learn), tiles resident; per-rank wall = host + launch chain + kernels + readback of its chunk")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=10)
    ap.add_argument("--cols", type=int, default=9)
    ap.add_argument("--tile", type=int, default=2048)
    ap.add_argument("--window", type=int, default=48)
    ap.add_argument("--method", default="surf", choices=["surf", "orb", "phase", "fuse", "surf_full"],
                    help="surf = the BASELINE metric; orb / phase time the other registration paths on the same grid; fuse = the"
                         " secondary metric of SURVEY 8d (mosaic assembly with fadeInAndFadeOut blending, N = 1 only)")
    ap.add_argument("--overlap", type=float, default=0.10, help="nominal tile overlap of the synthetic grid (SURVEY 8d: 10 %%)")
    ap.add_argument("--offset-evaluate", type=int, default=3, help="Method.offsetEvaluate (Main.py:12: 3)")
    ap.add_argument("--cpu-sample", type=int, default=12, help="pairs timed on the host cores for cpu_baseline (0 = skip)")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-resident-tiles measurement")
    ap.add_argument("--no-cold-leg", action="store_true", help="N = 1: skip the extra K steps without path memory (value_cold_path)")
    ap.add_argument("--no-path-memory", action="store_true", help="do not let the registrar use the scan pattern it learned from the previous step "
                    "(every step cold: history-driven speculation, blind chunk starts and chunks by pair count at N > 1)")
    ap.add_argument("--also-fuse", action="store_true", help="N = 1: after the registration line, time the mosaic assembly of the SAME resident tiles and print "
                    "its line too (a second JSON line; configs[4]: the 1024 tiles of 4096^2 are synthesised once for both)")
    ap.add_argument("--prior", default="other", choices=["other", "same"],
                    help="where the path memory of the timed steps is learned: other = a DIFFERENT instance of the scan pattern (same rows x cols x tile, "
                         "seed + 1: other texture, jitter and offsets -- the previous dataset of a session), registered once before the warm-up; "
                         "same = only the warm-up steps of the timed grid itself (round 4's headline)")
    ap.add_argument("--workload", default="grid", choices=["grid", "dendritic25"],
                    help="grid = the synthetic serpentine grid (BASELINE metric); dendritic25 = the 25 committed real pairs (N = 1, surf)")
    ap.add_argument("--from-files", action="store_true", help="N = 1: JPEG tiles on disk through Stitcher's ingest pipeline (decode inclusive)")
    ap.add_argument("--decode-threads", type=int, default=0, help="decoder threads of --from-files (0 = the Stitcher's default: one per host core, at most 32; 16 with VFSMS_NATIVE_JPEG=0)")
    ap.add_argument("--color", action="store_true", help="--from-files with colour JPEGs and isColorMode = True (Main.py:14's default)")
    ap.add_argument("--force-dist", action="store_true", help="N = 1: initialise the process group (nccl = RCCL) and run the step's all_gather through it "
                    "anyway -- the line then carries a non-null `collective` (RCCL start-up and the device-tensor all_gather exercised on one GPU)")
    ap.add_argument("--project-shards", default="", help="N = 1: after the timed steps, run the pair-sharded form of these rank counts (e.g. 2,4,8) "
                    "on THIS GPU, one emulated rank after the other (GridRegistrar.register_projected: the ranks' own code, each with the device to "
                    "itself), and add `projected_scaling` to the line: per-rank wall / GPU milliseconds, the fixed cost per step and the projected "
                    "pairs/s = pairs / (slowest rank + gather + assembly).  A projection from one GPU, not a measurement of N.")
    ap.add_argument("--all-tiles-on-every-rank", action="store_true", help="N > 1: keep the whole grid resident on every rank (rounds 1-5); default: a rank "
                    "synthesises, uploads and holds only the tiles of its chunk + one halo tile and re-fetches when the learned work split moves")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # VFSMS_DIST_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices, the
    # all-gather and the timing reduction run over gloo on host tensors); the driver's runs use the default, nccl == RCCL
    backend = os.environ.get("VFSMS_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world == 1:                                       # --force-dist without a launcher: a group of one, rendezvous on the loopback
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
        kw = dict(rank=rank, world_size=world)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **kw)
        else:
            dist.init_process_group(backend, **kw)
    coll_device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    assert world == args.gpus, "launch with --nproc-per-node == --gpus"
    if dist is not None:
        # fail loudly BEFORE anything is measured: the process group must span exactly --gpus ranks and, under RCCL, every rank must
        # sit on a GPU of its own (two ranks on one device would still produce a line -- of a job that is not the one asked for)
        assert dist.get_world_size() == args.gpus, "process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus)
        devs = [None] * world
        dist.all_gather_object(devs, (os.environ.get("HOSTNAME", ""), int(torch.cuda.current_device()), torch.cuda.get_device_properties(local_rank).name))
        if backend == "nccl":
            assert len({(h, d) for h, d, _n in devs}) == world, "ranks share a GPU under the nccl backend: %r" % (devs,)
            assert torch.cuda.device_count() >= world, "%d visible GPUs for %d ranks" % (torch.cuda.device_count(), world)

    import imagestitch_amd as isa
    from imagestitch_amd.grid import GridRegistrar
    from imagestitch_amd.distributed import make_all_gather, single_process_all_gather
    from imagestitch_amd.synthetic import SyntheticGrid

    eng = isa.Engine(local_rank)
    if args.workload == "dendritic25":
        return bench_dendritic25(args, eng, torch)
    if args.method == "surf_full":
        return bench_line_scan(args, eng, torch)
    grid = SyntheticGrid(args.rows, args.cols, args.tile, overlap=args.overlap)
    if args.from_files:
        return bench_from_files(args, eng, grid, torch)
    P = grid.n_pairs
    truth = np.array(grid.true_offsets(), np.int64)
    # Path memory (GridRegistrar.path_memory): the accepted directions of the previous registration of this scan pattern -- the warm-up
    # step -- are the speculation prior of the timed steps (batches planned over the whole predicted path from the first pair on; with
    # N > 1 one primed chain per rank instead of four blind ones, chunks cut by the predicted attempts).  Nothing comes from the ground
    # truth, every attempt is still evaluated, results never depend on it; --no-path-memory measures every step cold.
    bounds = GridRegistrar.chunk_bounds(P, world)
    lo, hi = bounds[rank]
    # a rank holds only the tiles of its chunk + one halo tile.  The chunk of the first (cold) step is cut by pair count; the
    # learned pattern moves the boundaries afterwards (chunks by predicted attempts), so the set is re-checked before every step
    # (ensure_tiles below) and the missing tiles are synthesised + uploaded then -- inside the timed region if it happens there.
    shard_local = world > 1 and not args.all_tiles_on_every_rank
    need = list(range(grid.n_tiles)) if (world > 1 and not shard_local) else (list(range(lo, hi + 1)) if hi > lo else [])
    # tiles live in pinned host memory (what a decoder feeding this engine would write into): uploads from it are asynchronous DMA
    tiles = {}
    # (grids beyond 128 tiles -- configs[4]: 1024 tiles of 4096^2, two core-hours of texture synthesis -- are generated by worker processes)
    t_start = time.perf_counter()
    gen_procs = 0 if grid.n_tiles <= 128 else max(2, min(96, (os.cpu_count() or 4) // (2 * world)))
    t_gen = time.perf_counter()
    for k, t in zip(need, grid.tiles(need, threads=min(8, os.cpu_count() or 1), processes=gen_procs)):
        buf = eng.pinned_empty(t.shape)
        buf[...] = t
        tiles[k] = buf
    t_gen = time.perf_counter() - t_gen
    big = grid.n_tiles > 128

    def progress(msg):
        if big and rank == 0:
            print("[bench %7.1f s] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)
    progress("%d tiles synthesised in %.1f s (%d worker processes)" % (len(need), t_gen, gen_procs))
    shapes = [(grid.th, grid.tw)] * grid.n_tiles
    handles = [None] * grid.n_tiles
    t_up = time.perf_counter()
    for k in need:
        handles[k] = eng.tile_upload(tiles[k])               # tiles resident in HBM before the timed region
    eng.sync()
    t_up = time.perf_counter() - t_up
    if args.method == "fuse":
        return bench_fuse(args, eng, grid, tiles, handles, torch)
    reg = GridRegistrar(eng, method=args.method, roiRatio=0.2, searchRatio=0.75, offsetEvaluate=args.offset_evaluate, directIncre=1,
                        surfParams=eng.surf_params() if args.method == "surf" else eng.orb_params() if args.method == "orb" else None,
                        window=args.window)
    reg.remember = not args.no_path_memory
    if os.environ.get("VFSMS_BENCH_PRIME", "0") not in ("", "0"):
        # PROFILING AID (tools/profile_round.sh, PMC passes of one step): start with the scan pattern already learned, so that every launch
        # of the short run is a steady-state launch.  Never set for a measured line.
        reg.path_memory = [int(d) for d in grid.true_directions()]
    gather = make_all_gather(coll_device) if dist is not None else single_process_all_gather

    refetch = dict(events=0, tiles=0, seconds=0.0, dropped=0)

    def ensure_tiles(g=None, hs=None, store=None):
        """shard-local: make the tiles of THIS rank's current chunk (+ halo) resident, release the ones the moved boundaries took away"""
        if not shard_local:
            return
        g = grid if g is None else g; hs = handles if hs is None else hs; store = tiles if store is None else store
        a, b = reg._bounds(P, world, None, reg._prediction(P, None), 1)[rank]
        want = set(range(a, b + 1)) if b > a else set()
        missing = sorted(k for k in want if hs[k] is None)
        if missing:
            t_f = time.perf_counter()
            for k, t in zip(missing, g.tiles(missing, threads=min(8, os.cpu_count() or 1), processes=gen_procs if len(missing) > 16 else 0)):
                if store is not None:
                    buf = eng.pinned_empty(t.shape); buf[...] = t; store[k] = buf
                hs[k] = eng.tile_upload(t)
            refetch["events"] += 1; refetch["tiles"] += len(missing); refetch["seconds"] += time.perf_counter() - t_f
        for k in range(len(hs)):
            if hs[k] is not None and k not in want:
                eng.tile_free(hs[k]); hs[k] = None
                if store is not None:
                    store.pop(k, None)
                refetch["dropped"] += 1

    def step(hs=None):
        if hs is None:
            ensure_tiles()
            hs = handles
        return reg.register_sharded(hs, shapes, 1, rank, world, gather)

    # The prior of the timed steps comes from ANOTHER dataset of the same scan pattern (what a session has: Main.py runs dataset after
    # dataset through one Stitcher): a second instance of the grid -- seed + 1: different texture, jitter, ground-truth offsets -- is
    # registered once, cold, and teaches the registrar the pattern; its tiles are released before the warm-up.  (--prior same: only the
    # warm-up steps of the timed grid teach it.)
    prior_note = "the warm-up steps of the timed grid itself (--prior same)"
    refetch_prior = dict(refetch)
    if args.prior == "other" and not args.no_path_memory and args.method in ("surf", "orb", "phase") and need:
        g2 = SyntheticGrid(args.rows, args.cols, args.tile, overlap=args.overlap, seed=grid.seed + 1)
        hs2 = [None] * grid.n_tiles
        for k, t in zip(need, g2.tiles(need, threads=min(8, os.cpu_count() or 1), processes=gen_procs)):
            hs2[k] = eng.tile_upload(t)
        res2, _d2 = step(hs2)                    # (cold: the chunks are those `need` was cut for)
        if args.method == "surf":
            t2 = np.array(g2.true_offsets(), np.int64)
            assert (res2[:, 0] == 1).all() and int(np.abs(res2[:, 1:3].astype(np.int64) - t2).max()) <= 1, "prior instance not registered"
        for k in need:
            eng.tile_free(hs2[k])
        refetch_prior = dict(refetch)
        assert reg.path_memory is not None and len(reg.path_memory) == P
        prior_note = ("a DIFFERENT instance of the scan pattern (same %d x %d x %d geometry, seed + 1: other texture, jitter and offsets), registered "
                      "once cold before the warm-up -- the previous dataset of a session" % (args.rows, args.cols, args.tile))

    def my_tiles():
        a, b = reg._bounds(P, world, None, reg._prediction(P, None), 1)[rank]
        return list(range(a, b + 1)) if b > a else []

    def step_from_host():
        """the same step with the tiles in host memory at its start: asynchronous uploads in path order on the copy stream (the
        first batch waits only for the tiles it names), registration, release of the device copies"""
        hs = [None] * grid.n_tiles
        ensure_tiles()                           # (shard-local: the pinned host copies of this rank's chunk)
        mine = my_tiles()
        for k in mine:
            hs[k] = eng.tile_upload_async(tiles[k])
        out = step(hs)
        for k in mine:
            eng.tile_free(hs[k])
        return out

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    progress("tiles resident (%.1f s of uploads); first step ..." % t_up)
    for _ in range(args.warmup):
        res, _d = step()
        progress("warm-up step done: %d attempts in %d batches so far, %d capacity retries" % (reg.stats["attempts"], reg.stats["batches"], getattr(reg, "capacity_retries", 0)))
    if args.warmup == 0:
        res, _d = step()
    t_w = time.perf_counter()                               # the clock of the steady-state warm-up starts AFTER the first (slow) steps
    # The MI355X needs about a second of sustained load to reach its steady clocks (and the first touches of the arena to
    # settle): measured here, the step right after a short warmup runs 5-100 % slower than the steady state.  More untimed
    # steps are run until the warmup has lasted MIN_WARM_S; they are reported, and the K timed steps below are exactly K.
    # The decision to run one more warm step is taken jointly (a step contains the collective): MAX over ranks of "not warm yet".
    warm_extra = 0
    while warm_extra < 400:
        more_warm = 1 if time.perf_counter() - t_w < MIN_WARM_S else 0
        if dist is not None:
            tw = torch.tensor([more_warm], dtype=torch.int64, device=coll_device)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            more_warm = int(tw.item())
        if not more_warm:
            break
        step()
        warm_extra += 1
    ok = res[:, 0] == 1
    err = np.abs(res[:, 1:3].astype(np.int64) - truth)
    max_err = int(err[ok].max()) if ok.any() else -1
    n_failed = int((~ok).sum())
    tol = 1 if args.method == "surf" else 0                  # SURF within 1 px, ORB exactly the ground truth
    # (phase: the reference as written adds cv2.phaseCorrelate's shift with the sign of the feature path, Stitcher.py:244-251 -- its offsets are
    #  not the ground truth by construction, see the line's note; not counted)
    off_truth = [int(k) for k in np.nonzero(ok & (err.max(axis=1) > tol))[0]] if args.method in ("surf", "orb") else None

    untimed = dict(reg.stats)                                # what the process registered before the timed steps (prior instance, warm-up)
    for k in reg.stats:
        reg.stats[k] = 0
    eng.profile_enable(True)
    eng.profile_read(reset=True)
    fence()
    refetch_at_t0 = dict(refetch)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    progress("%d timed steps: %.1f ms per step" % (args.steps, elapsed / args.steps * 1e3))
    prof = eng.profile_read(reset=True)
    eng.profile_enable(False)
    st = dict(reg.stats)

    elapsed_host = None
    if not args.no_host_leg:
        res_h, _d = step_from_host()                              # warm the tile pool
        assert np.array_equal(res_h, res)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_from_host()
        fence()
        elapsed_host = time.perf_counter() - t0
    elapsed_cold = None
    if world == 1 and not args.no_path_memory and not args.no_cold_leg:
        reg_c = GridRegistrar(eng, method=args.method, roiRatio=0.2, searchRatio=0.75, offsetEvaluate=args.offset_evaluate, directIncre=1,
                              surfParams=reg.params, window=args.window)
        reg_c.remember = False
        res_c, _d = reg_c.register_sharded(handles, shapes, 1, 0, 1, single_process_all_gather)
        assert np.array_equal(res_c, res)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            reg_c.register_sharded(handles, shapes, 1, 0, 1, single_process_all_gather)
        fence()
        elapsed_cold = time.perf_counter() - t0
        cold_stats = dict(reg_c.stats)
    projected = None
    if world == 1 and args.project_shards and args.method in ("surf", "orb", "phase"):
        projected = project_shards(args, eng, reg, handles, shapes, res, P, elapsed / args.steps, fence, gather if dist is not None else None, progress)
    per_rank = None
    if dist is not None:
        lo_t, hi_t = reg._bounds(P, world, None, reg._prediction(P, None), 1)[rank]      # the chunk of the timed steps (cut by the predicted attempts)
        mine = dict(rank=rank, pairs=hi_t - lo_t, attempts_per_step=st["attempts"] / max(args.steps, 1), batches_per_step=st["batches"] / max(args.steps, 1),
                    gpu_ms_per_step=round(sum(v[0] for v in prof.values()) / max(args.steps, 1), 3), wall_ms_per_step=round(elapsed / args.steps * 1e3, 3),
                    tiles_resident=sum(1 for h in handles if h is not None), tiles_synthesised_first=len(need),
                    tiles_refetched_after_the_split_moved=refetch["tiles"] - refetch_prior["tiles"], refetch_s=round(refetch["seconds"] - refetch_prior["seconds"], 3),
                    refetches_inside_timed_steps=refetch["events"] - refetch_at_t0["events"])
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        t = torch.tensor([elapsed, elapsed_host or 0.0], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        elapsed_host = float(t[1].item()) if elapsed_host is not None else None

    # ---- rooflines from the live HIP-event timings of this rank (library-side events on the kernels' own stream) ----
    stages = {k: dict(ms=round(v[0], 3), launches=v[1], ms_per_launch=round(v[0] / max(v[1], 1), 4)) for k, v in prof.items()}
    roofline = None
    extra = {}
    roi_h, roi_w = isa.roi_rect((grid.th, grid.tw), 1, "first", 0.2)[2:]
    # A "launch" of the rooflines below is one launch GROUP = one fused batch of the registrar.  Since round 5 a large batch is cut in two parts
    # (the 2-NN search of part 0 runs on a second stream beside the detect stage of part 1, csrc/api.hip: attempt_surf_impl), so a stage may be
    # enqueued twice per batch and the stages of the second stream carry "@s2": times are summed per stage, divided by the batches.
    groups = int(st["batches"])

    second_stream = any(k.endswith("@s2") for k in prof)

    def stage(name):
        a, b = prof.get(name, (0.0, 0)), prof.get(name + "@s2", (0.0, 0))
        if not second_stream:
            return (a[0], a[1])                       # the profiled launch groups themselves (a capacity retry is a group of its own)
        return (a[0] + b[0], groups if (a[1] + b[1]) else 0)
    de_ms, de_n = stage("describe")
    if de_n and args.method == "surf":
        # dominant kernels: k_describe + k_describe_small (descriptor windows).  Until the row-pair image they were bound by the
        # texture-address path (TA busy 73 % of the launch: two gathers per sample); with one gather per sample the VALU is the
        # busiest unit (PMC: SQ_INSTS_VALU x 4 cycles = ~94 % of the SIMD cycles of k_describe, TA ~60 %); HBM traffic is a few per
        # cent of what 8 TB/s would move in that time.
        # Algorithmic work per launch = bilinear samples (win x win per keypoint, win = int(21 * size * 1.2 / 9)) x the VALU
        # instructions one sample takes in the kernel's own inner loop; samples/keypoint is measured outside the timed region.
        k0 = min(tiles) if isinstance(tiles, dict) and tiles else need[0]
        ra = isa.roi_rect((grid.th, grid.tw), 1, "first", 0.2)
        _k, _d, kf = eng.surf_detect_describe(np.ascontiguousarray(tiles[k0][ra[0]:ra[0] + ra[2], ra[1]:ra[1] + ra[3]]), full=True)
        win = np.minimum((21 * (kf["size"] * np.float32(1.2) / np.float32(9.0))).astype(np.int64), 739)
        spk = float((win.astype(np.float64) ** 2).mean()) if len(win) else 0.0
        kps = st["sum_nq_plus_nt"] / de_n                                   # keypoints described per launch
        dur = de_ms / de_n * 1e-3
        laneops = kps * spk * DESC_OPS_LOWER_BOUND
        traffic, traffic_src = pmc_traffic_scaled("k_describe", st["attempts"] / de_n)
        traffic_small, _src_s = pmc_traffic_scaled("k_describe_small", st["attempts"] / de_n)
        traffic = (traffic + traffic_small) if (traffic is not None and traffic_small is not None) else None     # the same kernels as compulsory_bytes
        valu_insts, _src = pmc_value("k_describe", "INSTS_VALU")
        busy, _src2 = pmc_value("k_describe", "BUSY_CYCLES")            # summed over the 32 shader engines: / 32 = cycles of the launch
        valu_busy = round(valu_insts * 4.0 / (busy / 32.0 * 1024.0), 3) if valu_insts and busy else None
        roofline = dict(kernel="k_describe+k_describe_small", bound="valu", achieved=round(laneops / dur / 1e12, 3), peak=round(VALU_PEAK_TLANEOPS, 2),
                        unit="Tlane-op/s", frac=round(laneops / dur / 1e12 / VALU_PEAK_TLANEOPS, 4), traffic=traffic, traffic_source=traffic_src,
                        # the other reading of the VALU roof: MI355X_MICROARCH.md's "wave64 VALU instruction over 2 cycles" / two elements per packed
                        # f32 instruction = 78.6 T lane-ops/s; only plain VOP2 instructions get near it here (profiles/r05_valu_peak.txt: v_mul_f32
                        # 64.9, v_add_u32 54 T), the kernel's f64 / conversion / VOP3 mix issues in the 4-cycle class.  Both are printed.
                        peak_2cycle_or_packed=round(2 * VALU_PEAK_TLANEOPS, 2), frac_of_2cycle_or_packed_peak=round(laneops / dur / 1e12 / (2 * VALU_PEAK_TLANEOPS), 4),
                        traffic_over_compulsory=(round(traffic / max(kps / max(len(kf), 1) * (2.0 * roi_h * roi_w) + kps * 441.0, 1.0), 3) if traffic else None),
                        compulsory_bytes_per_launch=round(kps / max(len(kf), 1) * (2.0 * roi_h * roi_w) + kps * 441.0), avg_launch_ms=round(dur * 1e3, 4),
                        lane_ops_per_launch=laneops, kernel_valu_insts_per_sample_inner_loop=DESC_VALU_PER_SAMPLE, keypoints_per_launch=kps,
                        samples_per_keypoint=round(spk, 1), launches=de_n,
                        note="dominant stage by time (%.0f %% of the GPU time of a step; the timed scope also holds k_pair_rows, k_desc_order, "
                             "k_desc_plan, k_desc_recs and k_desc_tail); achieved = bilinear samples x the 26 lane-operations the REFERENCE's expression "
                             "needs per sample (ops_per_sample_lower_bound: independent of how the kernel is written; rounds 1-4 counted the kernel's own "
                             "30 instructions per sample) / the live HIP-event duration; all VALU instructions k_describe issues (PMC SQ_INSTS_VALU, "
                             "profiles/) keep its SIMDs busy for valu_busy_frac_pmc of the launch (bookkeeping of the staging units, border units, "
                             "INTER_AREA reduction, row-origin chains)"
                             % (100.0 * de_ms / max(sum(v[0] for v in prof.values()), 1e-9)),
                        valu_insts_per_launch_pmc=valu_insts, valu_busy_frac_pmc=valu_busy, ta_busy_frac_pmc=pmc_value("k_describe", "ta_busy_frac")[0],
                        valu_peak_source="profiles/r05_valu_peak.txt (tools/valu_peak.hip on the MI355X box: 4-cycle class instructions 33-38 T lane-ops/s)",
                        ops_per_sample_lower_bound=DESC_OPS_LOWER_BOUND,
                        **clock_fields(laneops / dur / 1e12),
                        valu_insts_lower_bound_per_launch=round(kps * spk * DESC_OPS_LOWER_BOUND / 64.0),
                        valu_issued_over_lower_bound=valu_over_bound(spk),
                        per_launch_pmc_note="PMC figures are per launch OF THE PMC RUN (its launches need not have this run's size); the ratio "
                                            "valu_issued_over_lower_bound is formed from counter TOTALS and the keypoints that run described (pmc_run line)")
    bf_ms, bf_n = stage("bf_mfma")
    if bf_n:
        dur = bf_ms / bf_n * 1e-3
        flops = 2.0 * 64 * st["sum_nq_nt"] / bf_n                         # one 64-d dot product per (query, train)
        bytes_ = (st["sum_nq_plus_nt"] * 64 * 4) / bf_n
        f32_filter = os.environ.get("VFSMS_BF_F32FILTER", "0") not in ("", "0")
        peak = FP32_PEAK_TFLOPS if f32_filter else BF16_PEAK_TFLOPS
        kname = "k_bf_mfma_d64" if f32_filter else "k_bf_split16+k_bf_mfma16_d64"
        extra["bf_l2_mfma"] = dict(kernel=kname, bound="mfma", achieved=round(flops / dur / 1e12, 3), peak=peak,
                                   unit="TFLOP/s", frac=round(flops / dur / 1e12 / peak, 4),
                                   traffic=(pmc_traffic("k_bf_mfma_d64")[0] if f32_filter else
                                            sum_or_none([pmc_traffic_scaled("void k_bf_mfma16_d64<%d>" % q, st["attempts"] / bf_n)[0] for q in (0, 1)])),
                                   avg_launch_ms=round(dur * 1e3, 4), flops_per_launch=flops, launches=bf_n,
                                   issued_mfma_flops_per_launch=flops if f32_filter else flops * 4.5,
                                   note=("v_mfma_f32_32x32x2_f32 candidate filter (peak = dense f32 MFMA)" if f32_filter else
                                         "split-bf16 candidate filter in two sweeps: bounds from hi.hi (5 MFMA k-steps per tile), then q.t = hi.hi + hi.lo + lo.hi "
                                         "(13) on v_mfma_f32_32x32x16_bf16 -- 18 issued k-steps per 4 algorithmic ones = 4.5x the flops; peak = dense "
                                         "bf16 MFMA; achieved counts one 64-d dot product per pair") +
                                        "; the exact distances are evaluated by k_bf_verify_d64 for the few surviving candidates")
        extra["bf_l2_hbm"] = dict(bound="hbm", achieved=round(bytes_ / dur / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                                  frac=round(bytes_ / dur / 1e9 / HBM_PEAK_GBS, 5), bytes_per_launch=bytes_)
    in_ms, in_n = stage("integral")
    if in_n:
        # algorithmic bytes of cv::integral per ROI: h*w (u8 in) + 4 (h+1)(w+1) (i32 out) ~ 5 B/px
        extra["integral_hbm"] = hbm_roofline("k_integral_final+k_integral_bandsum+k_integral_bandscan", st["roi_px"] / in_n * 5.0, in_ms / in_n, in_n)
        tr = [pmc_traffic_scaled(k, st["attempts"] / in_n)[0] for k in ("k_integral_final", "k_integral_bandsum", "k_integral_bandscan")]
        extra["integral_hbm"]["traffic"] = sum(tr) if all(v is not None for v in tr) else None
    he_ms, he_n = stage("hessian")
    if he_n:
        # bytes the stage moves as it is built: the integral image read once per octave pass, 4 (h+1)(w+1) x 4 = 16 B/px, + the determinant
        # layers written, 4 x sum_o 5 (h/2^o)(w/2^o) = 26.6 B/px -- 42.6 B/px.  (SURVEY 8d's 69.1 B/px also counts the trace layers, which
        # have not been written since round 3: the sign of the Laplacian is recomputed for the few thousand candidates.)
        HESS_BYTES_PER_PX = 16.0 + 4.0 * 5.0 * (1 + 1 / 4.0 + 1 / 16.0 + 1 / 64.0)
        HK = ("void k_hessian_lds<1, 64, 4>", "void k_hessian_lds<2, 32, 8>", "k_hessian_rows2", "k_hessian_coarse")      # (names as the PMC summary cuts them: 28 characters)
        extra["hessian_hbm"] = hbm_roofline("k_hessian_lds<1,64,4>+k_hessian_lds<2,32,8>+k_hessian_rows2(octave 2)+k_hessian_coarse(octave 3)",
                                            st["roi_px"] / he_n * HESS_BYTES_PER_PX, he_ms / he_n, he_n)
        extra["hessian_hbm"]["bytes_per_px"] = round(HESS_BYTES_PER_PX, 2)
        hv, _s = pmc_value(HK[0], "INSTS_VALU"); hb, _s = pmc_value(HK[0], "BUSY_CYCLES")
        extra["hessian_hbm"]["valu_insts_x4_over_simd_cycles_octave0"] = round(hv * 4.0 / (hb / 32.0 * 1024.0), 3) if hv and hb else None
        htr = [pmc_traffic_scaled(k, st["attempts"] / he_n) for k in HK]
        extra["hessian_hbm"]["traffic"] = sum(v[0] for v in htr) if all(v[0] is not None for v in htr) else None
        extra["hessian_hbm"]["traffic_source"] = htr[0][1]
        extra["hessian_hbm"]["note"] = ("bytes = what the stage moves (42.6 B/px: no trace layers; SURVEY 8d's 69.1 B/px counted them); the fine octaves are bound by VALU "
                                        "issue + LDS taps, not by HBM: SQ_INSTS_VALU x 4 cycles EXCEEDS the SIMD cycles of octave 0's launches "
                                        "(valu_insts_x4_over_simd_cycles_octave0 > 1) -- part of its instructions are plain VOP2 integer adds, which issue in "
                                        "~2.7 cycles (profiles/r05_valu_peak.txt): the kernel has no idle issue slots")
    if args.method == "phase":
        ph_ms, ph_n = prof.get("phase", (0.0, 0))
        if ph_n:
            M, N = optimal_dft_size(roi_h), optimal_dft_size(roi_w)
            per_attempt = 2 * roi_h * roi_w + 240 * M * (N // 2 + 1) + 8 * M * N
            plan = eng.phase_plan(roi_h, roi_w)
            if plan["lds_transforms"]:
                # since round 6 the transforms run in LDS: a strip's spectra cross HBM three times (rows -> columns -> rows), not the 2 x 2 x 3 passes
                # + 3 cross-power streams SURVEY 8d's figure counts for a transform library -- `achieved` stays on SURVEY's bytes (the work done),
                # `bytes_moved_by_this_design` is what these kernels read and write
                Mc, Nr = plan["M"], plan["N"]
                hh = roi_w if plan["transposed"] else roi_h
                Ct = plan["columns_per_workgroup"]
                ncp = ((Nr // 2 + 1 + Ct - 1) // Ct) * Ct
                moved = (2 * roi_h * roi_w * (3 if plan["transposed"] else 1) + 2 * 2 * 16 * hh * (Nr // 2 + 1) + 2 * 16 * Mc * ncp + 8 * Mc * Nr)
                roofline = hbm_roofline("k_phase_rows_fwd + k_phase_cols + k_phase_rows_inv + k_peak_centroid", per_attempt * st["attempts"] / ph_n, ph_ms / ph_n, ph_n,
                                        note="one 'launch' = one batched phase correlation over all attempts of a batch: FP64 row transforms (packed real, LDS), "
                                             "column transforms + cross power + inverse column transforms in ONE kernel (LDS), inverse rows + arg-max, centroid; "
                                             "bytes per attempt = SURVEY 8d's 2hw + 240 M (N/2+1) + 8 MN (a transform library's passes); this design moves "
                                             "bytes_moved_by_this_design per attempt, and its kernels are 50-70 % VALU-busy (FP64 butterflies + addressing: "
                                             "profiles/r06_pmc_phase.txt) -- no unit of the chip is the single roof",
                                        attempts_per_launch=st["attempts"] / ph_n, padded=[M, N], plan=plan,
                                        bytes_moved_by_this_design=moved, frac_of_bytes_moved=round(moved * st["attempts"] / ph_n / (ph_ms / ph_n * 1e-3) / 8e12, 4))
                # PMC traffic of the path's kernels (profiles/*_pmc_summary.txt, phase section: batches of 32 attempts) per attempt x this run's attempts per launch
                parts = [pmc_total(kname, "total(x2 rule)")[0] for kname in ("k_phase_rows_fwd", "k_phase_cols", "k_phase_rows_inv", "k_peak_centroid")]
                tot = sum(parts) + (pmc_total("k_phase_transpose_u8", "total(x2 rule)")[0] or 0.0) if all(v is not None for v in parts) else None
                _t, n_cols = pmc_total("k_phase_cols", "total(x2 rule)")
                apl, src = pmc_value("k_phase_cols", "attempts_per_launch")
                if tot and n_cols and apl:
                    roofline["traffic"] = tot / (n_cols * apl) * st["attempts"] / ph_n
                    roofline["traffic_source"] = src
                    roofline["traffic_over_bytes_moved"] = round(roofline["traffic"] / (moved * st["attempts"] / ph_n), 3)
                    roofline["valu_busy_frac_pmc"] = {k: pmc_value(k, "valu_busy_frac")[0] for k in ("k_phase_rows_fwd", "k_phase_cols", "k_phase_rows_inv")}
            else:
                roofline = hbm_roofline("rocFFT r2c/c2r + k_pad_u8_f64 + k_cross_power + k_argmax_*", per_attempt * st["attempts"] / ph_n, ph_ms / ph_n, ph_n,
                                        note="one 'launch' = one batched phase correlation (pad, 2 forward + 1 inverse FP64 transforms, cross power, "
                                             "argmax, centroid) over all attempts of a batch; bytes per attempt = 2hw + 240 M (N/2+1) + 8 MN",
                                        attempts_per_launch=st["attempts"] / ph_n, padded=[M, N], plan=plan)
    if args.method == "orb":
        fa_ms, fa_n = prof.get("orb_fast", (0.0, 0))
        if fa_n:
            # fused FAST-9/16 + NMS over the 8-level pyramid of every ROI: each level read once (sum of level areas = 3.27 h w at
            # scale 1.2), 1 B/px; scores never leave LDS
            longest = max(prof.items(), key=lambda kv: kv[1][0]) if prof else (None, (0.0, 0))
            roofline = hbm_roofline("k_orb_fast_nms", st["roi_px"] / fa_n * 3.27, fa_ms / fa_n, fa_n,
                                    note="the longest single KERNEL of the ORB path and its one streaming pass (bytes = pyramid levels read once); the "
                                         "longest STAGE is `longest_stage` -- orb_select is six kernels of per-keypoint work (Harris, rank, angle), "
                                         "latency-bound, with no bytes-per-pixel figure to hold against HBM",
                                    longest_stage=longest[0], longest_stage_ms_per_launch=round(longest[1][0] / max(longest[1][1], 1), 4))
        py_ms, py_n = prof.get("orb_pyramid", (0.0, 0))
        if py_n:
            extra["orb_pyramid_hbm"] = hbm_roofline("k_orb_resize", st["roi_px"] / py_n * (3.27 + 2.27), py_ms / py_n, py_n)
        bh_ms, bh_n = prof.get("bf_hamming", (0.0, 0))
        if bh_n:
            extra["bf_hamming_ms_per_launch"] = round(bh_ms / bh_n, 4)

    # ---- CPU baseline: the oracle (a port, cv2 is not installable) on a bounded sample, rank 0 at N = 1 only ------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0 and args.method == "surf":
        cpu = cpu_baseline_surf(args, grid, tiles, isa)
    elif rank == 0 and world == 1 and args.cpu_sample > 0 and args.method in ("orb", "phase"):
        cpu = cpu_baseline_pairs(args, grid, tiles, isa, args.method)

    if rank == 0:
        pmc_info = pmc_build()
        if roofline is not None:
            roofline["pmc_stale"] = pmc_info["pmc_stale"]
        out = {
            "metric": "image-pairs/sec (%dx%d grayscale, SURF+BF)" % (args.tile, args.tile) if args.method == "surf" else "image-pairs/sec (%s)" % args.method,
            "value": round(P * args.steps / elapsed, 3),
            "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_extra_steps_until_1p5s": warm_extra,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if args.method == "phase" else "u8" if args.method == "orb" else "f32",
            "data": "synthetic",
            "config": {"workload": "synthetic %dx%d grid of %dx%d u8 tiles (overlap %.0f %%), serpentine path, %d pairs; %s; roiRatio 0.2, offsetEvaluate %d, "
                                   "direction 1, directIncre 1" % (args.rows, args.cols, args.tile, args.tile, 100 * args.overlap, P,
                                      {"surf": "SURF(100,4,3,64-d)+BF-L2 knn2 ratio 0.75 + mode vote", "orb": "ORB(5000,1.2,8)+BF-Hamming 1-NN + mode vote",
                                       "phase": "FFT phase correlation of the ROI strips"}[args.method], args.offset_evaluate),
                       "pairs": P, "parallelism": "pairs%d" % world, "speculation_window": args.window,
                       "path_prediction": ("none (--no-path-memory): every step registers the path cold" if args.no_path_memory else
                                           "path memory: the accepted directions of the previous registration of this scan pattern drive the speculation plan "
                                           "of the timed steps; nothing from the ground truth, every attempt evaluated; first learned from " + prior_note)},
            "max_abs_offset_error_px": max_err, "pairs_failed": n_failed,
            "pairs_off_truth": len(off_truth) if off_truth is not None else None, "pairs_off_truth_indices": off_truth[:40] if off_truth is not None else None,
            "pairs_off_truth_note": ("%d of %d pairs accepted off truth (tolerance %d px)%s" % (
                len(off_truth), P, tol, "; reference behaviour at %d votes: every ORB query votes, ImageUtility.py:297-302" % args.offset_evaluate
                if args.method == "orb" and off_truth else "")) if off_truth is not None else None,
            "path_memory_primed_for_profiling": os.environ.get("VFSMS_BENCH_PRIME", "0") not in ("", "0"),
            "value_cold_path": round(P * args.steps / elapsed_cold, 3) if elapsed_cold else None,
            "cold_path": (dict(ms_per_step=round(elapsed_cold / args.steps * 1e3, 3), attempts_per_step=cold_stats["attempts"] / (args.steps + 1),
                               batches_per_step=cold_stats["batches"] / (args.steps + 1),
                               note="the same K steps by a registrar WITHOUT path memory: every path learned from scratch (history-driven speculation), "
                                    "rounds 1-3's headline configuration") if elapsed_cold else None),
            "value_host_resident_tiles": round(P * args.steps / elapsed_host, 3) if elapsed_host else None,
            "ms_per_step_host_resident_tiles": round(elapsed_host / args.steps * 1e3, 3) if elapsed_host else None,
            "h2d_ms_rank0_blocking": round(t_up * 1e3, 2), "tile_synthesis_s": round(t_gen, 1),
            "attempts_per_step": st["attempts"] / max(args.steps, 1), "batches_per_step": st["batches"] / max(args.steps, 1),
            "keypoints_per_roi": round(st["sum_nq_plus_nt"] / max(2 * st["attempts"], 1), 1) if args.method == "surf" else None,
            # everything this process registered up to the end of the timed steps (rank 0): what a PMC pass over the whole run has counted
            "process_totals_through_timed_steps": dict(attempts=untimed["attempts"] + st["attempts"], keypoints=untimed["sum_nq_plus_nt"] + st["sum_nq_plus_nt"]),
            "capacity_retries": getattr(reg, "capacity_retries", 0),
            "roofline": roofline,
            "pmc": pmc_info,
            "cpu_baseline": cpu,
            "stages": stages,
            "stages_sum_ms_per_step": round(sum(v[0] for v in prof.values()) / max(args.steps, 1), 3),
            "stages_second_stream_ms_per_step": round(sum(v[0] for k, v in prof.items() if k.endswith("@s2")) / max(args.steps, 1), 3),
            "overlap_note": ("stages named @s2 were enqueued on the second compute stream (2-NN search + vote of the first part of a batch beside the detect "
                             "stage of the second part): ms_per_step < stages_sum_ms_per_step by what the two pipes hid of each other") if second_stream else None,
            "per_rank": per_rank,
            "tiles_per_rank": ("shard-local: chunk + one halo tile, re-fetched when the learned split moves (per_rank[].tiles_*)" if shard_local else
                               "all tiles on every rank" if world > 1 else "one rank"),
            "projected_scaling": projected,
            "collective": (dict(backend=dist.get_backend(), world_size=dist.get_world_size(), device=str(coll_device), rank_devices=devs,
                                prediction_repair_rounds=getattr(reg, "hint_repairs", 0),
                                op="one all_gather of the int32 offset tables per step") if dist is not None else None),
        }
        if args.method == "orb" and max_err > 1:
            out["note"] = ("max_abs_offset_error_px is the reference's own ORB search at offsetEvaluate = %d: every query votes (no ratio test, no "
                           "distance threshold, ImageUtility.py:297-302), so a wrong candidate direction now and then collects %d equal votes and "
                           "is ACCEPTED (on the reference's real tiles too: tests/golden/dendritic_path_oracle_orb.json, 15 of 87 pairs); the engine "
                           "takes the oracle's decisions one for one (tests: test_orb_grid_at_offset_evaluate_3_equals_oracle_chain); with "
                           "--offset-evaluate 10 every offset equals the ground truth" % (args.offset_evaluate, args.offset_evaluate))
        if args.method == "phase" and max_err > 1:
            out["note"] = ("max_abs_offset_error_px is the reference as written: cv2.phaseCorrelate returns the shift of B's content relative to A's "
                           "and Stitcher.py:244-251 adds it with the sign of the feature path (SURVEY 8a-G; configs[0] gives [1400, 0] where the "
                           "true offset is ~[1699, -1]); parity target = the reference's arithmetic, phaseSignFix is the opt-in correction")
        out.update(extra)
        print(json.dumps(out), flush=True)
    if args.also_fuse and world == 1:
        progress("mosaic assembly ...")
        bench_fuse(args, eng, grid, tiles, handles, torch, close=False)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()

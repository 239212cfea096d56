#!/usr/bin/env python
"""bench.py -- image-pairs/sec of the VFSMS registration hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric / SURVEY section 8d): a synthetic 10 x 9 grid of 2048 x 2048 grayscale tiles on a
column-major serpentine path (89 consecutive pairs, 8 turns), SURF (hessian 100, 4 octaves, 3 layers, 64-d) +
BF-L2 2-NN + ratio 0.75 + mode vote (>= 3), incremental ROI (roiRatio 0.2) with direction rotation
(direction 1, directIncre 1) -- exactly Main.py's settings.  One "step" = registering all 89 pairs, tiles
already resident in HBM.  With N > 1 the pairs are sharded in contiguous chunks, one process per GPU, and ONE
all-gather (RCCL) of the int32 offset tables closes the step ("strong" scaling: total work is fixed).
Prints ONE JSON line on rank 0.
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
FP32_PEAK_TFLOPS = 157.3       # dense FP32 MFMA peak of gfx950 (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_pmc_summary.txt, written by
    tools/profile_round.sh from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command; gfx950 x2
    correction applied to FETCH_SIZE).  PMC counters cannot be collected from inside the timed run, hence the file."""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.txt")), reverse=True):
        for line in open(path):
            m = re.match(r"^%s\s+launches=\d+ .*total\(x2 rule\)=([0-9.e+]+) B/launch" % re.escape(kernel), line)
            if m:
                return float(m.group(1)), os.path.relpath(path, ROOT)
    return None, None


def bench_fuse(args, eng, grid, tiles, handles, torch):
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

    def assemble(download, resident=True):
        canvas = eng.canvas_create(rows, cols, 1)
        try:
            for i in range(n):
                oy, ox = offsetList[i]
                if i == 0:
                    eng.canvas_paste_tile(canvas, handles[i], oy, ox) if resident else eng.canvas_paste(canvas, tiles[i], oy, ox)
                    continue
                roi = (max(oy, rangeX[i - 1][0]), max(ox, rangeY[i - 1][0]),
                       min(oy + grid.th, rangeX[i - 1][1]), min(ox + grid.tw, rangeY[i - 1][1]))
                if resident:
                    eng.canvas_fuse_tile_resident(canvas, handles[i], oy, ox, roi, offs[i][0], offs[i][1])
                else:
                    eng.canvas_fuse_tile(canvas, tiles[i], oy, ox, roi, offs[i][0], offs[i][1])
            eng.sync()
            return eng.canvas_download(canvas, rows, cols, 1) if download else None
        finally:
            eng.canvas_free(canvas)

    for _ in range(max(args.warmup, 1)):
        assemble(False)
    torch.cuda.synchronize(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        assemble(False)
    torch.cuda.synchronize(); eng.sync()
    dt = (time.perf_counter() - t0) / args.steps
    t1 = time.perf_counter()
    out = assemble(True)
    dl = time.perf_counter() - t1 - dt
    t2 = time.perf_counter()
    out_host = assemble(True, resident=False)
    dt_host = time.perf_counter() - t2 - dl
    assert np.array_equal(out, out_host)
    mpx = rows * cols / 1e6
    print(json.dumps({
        "metric": "fuse Mpx/sec (mosaic pixels, fadeInAndFadeOut)", "value": round(mpx / dt, 2), "unit": "Mpx/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "mosaic of the synthetic %dx%d grid of %dx%d u8 tiles from its true offsets: canvas %d x %d"
                               % (args.rows, args.cols, args.tile, args.tile, rows, cols), "tiles": n, "tiles_resident_in_hbm": True,
                   "canvas_download_ms": round(dl * 1e3, 1), "ms_per_step_with_host_tiles": round(dt_host * 1e3, 1)},
        "tile_Mpx_per_s": round(n * grid.th * grid.tw / 1e6 / dt, 2), "mosaic_nonzero_fraction": round(float((out > 0).mean()), 4)}))
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=10)
    ap.add_argument("--cols", type=int, default=9)
    ap.add_argument("--tile", type=int, default=2048)
    ap.add_argument("--window", type=int, default=24)
    ap.add_argument("--method", default="surf", choices=["surf", "orb", "phase", "fuse"],
                    help="surf = the BASELINE metric; orb / phase time the other registration paths on the same grid; fuse = the"
                         " secondary metric of SURVEY 8d (mosaic assembly with fadeInAndFadeOut blending, N = 1 only)")
    ap.add_argument("--cpu-sample", type=int, default=6, help="pairs timed on the host cores for cpu_baseline (0 = skip)")
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
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    coll_device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    assert world == args.gpus, "launch with --nproc-per-node == --gpus"

    import imagestitch_amd as isa
    from imagestitch_amd.grid import GridRegistrar
    from imagestitch_amd.distributed import make_all_gather, single_process_all_gather
    from imagestitch_amd.synthetic import SyntheticGrid

    eng = isa.Engine(local_rank)
    grid = SyntheticGrid(args.rows, args.cols, args.tile, overlap=0.10 if args.method == "surf" else 0.15)
    P = grid.n_pairs
    truth = np.array(grid.true_offsets(), np.int64)
    bounds = GridRegistrar.chunk_bounds(P, world)
    lo, hi = bounds[rank]
    need = list(range(lo, hi + 1)) if hi > lo else []
    tiles = dict(zip(need, grid.tiles(need, threads=min(8, os.cpu_count() or 1))))
    shapes = [(grid.th, grid.tw)] * grid.n_tiles
    handles = [None] * grid.n_tiles
    t_up = time.perf_counter()
    for k in need:
        handles[k] = eng.tile_upload(tiles[k])               # tiles resident in HBM before the timed region
    eng.sync()
    t_up = time.perf_counter() - t_up                        # one H2D copy per tile (pageable host memory), reported as a side note only
    if args.method == "fuse":
        return bench_fuse(args, eng, grid, tiles, handles, torch)
    reg = GridRegistrar(eng, method=args.method, roiRatio=0.2, searchRatio=0.75, offsetEvaluate=3 if args.method == "surf" else 10, directIncre=1,
                        surfParams=eng.surf_params() if args.method == "surf" else eng.orb_params() if args.method == "orb" else None,
                        window=args.window)
    gather = make_all_gather(coll_device) if world > 1 else single_process_all_gather

    def step():
        return reg.register_sharded(handles, shapes, 1, rank, world, gather)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    for _ in range(args.warmup):
        res, _d = step()
    if args.warmup == 0:
        res, _d = step()
    ok = res[:, 0] == 1
    err = np.abs(res[:, 1:3].astype(np.int64) - truth)
    max_err = int(err[ok].max()) if ok.any() else -1
    n_failed = int((~ok).sum())

    for k in reg.stats:
        reg.stats[k] = 0
    eng.profile_enable(True)
    eng.profile_read(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read(reset=True)
    eng.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- rooflines from the live HIP-event timings of this rank (library-side events on the kernels' own stream) ----
    stages = {k: dict(ms=round(v[0], 3), launches=v[1], ms_per_launch=round(v[0] / max(v[1], 1), 4)) for k, v in prof.items()}
    st = reg.stats
    roofline = None
    extra = {}
    de_ms, de_n = prof.get("describe", (0.0, 0))
    if de_n and args.method == "surf":
        # dominant kernel: k_describe (descriptor windows).  Algorithmic traffic per keypoint (SURVEY 8d / DESIGN.md 5): the
        # win x win bilinear samples of the rotated window, 4 source bytes each, plus the 21 x 21 patch written out.
        # samples/keypoint is measured outside the timed region on the first ROI pair (win = int(21 * size * 1.2 / 9)).
        ra = isa.roi_rect((grid.th, grid.tw), 1, "first", 0.2)
        k0 = need[0]
        _k, _d, kf = eng.surf_detect_describe(np.ascontiguousarray(tiles[k0][ra[0]:ra[0] + ra[2], ra[1]:ra[1] + ra[3]]), full=True)
        win = np.minimum((21 * (kf["size"] * np.float32(1.2) / np.float32(9.0))).astype(np.int64), 739)
        spk = float((win.astype(np.float64) ** 2).mean()) if len(win) else 0.0
        kps = st["sum_nq_plus_nt"] / de_n                                   # keypoints described per launch
        b = kps * (4.0 * spk + 441.0)
        dur = de_ms / de_n * 1e-3
        traffic, traffic_src = pmc_traffic("k_describe")
        roofline = dict(kernel="k_describe", bound="hbm", achieved=round(b / dur / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(b / dur / 1e9 / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_src, avg_launch_ms=round(dur * 1e3, 4),
                        bytes_per_launch=b, keypoints_per_launch=kps, samples_per_keypoint=round(spk, 1), launches=de_n,
                        note="dominant kernel by time; it is VALU-issue bound, not HBM bound: ~45 instructions per bilinear sample "
                             "(double-precision sample positions as in the reference), PMC in profiles/ shows the SIMDs issuing >90 % "
                             "of cycles; HBM traffic measured by PMC is a few tens of MB per launch")
    bf_ms, bf_n = prof.get("bf_mfma", (0.0, 0))
    if bf_n:
        dur = bf_ms / bf_n * 1e-3
        flops = 2.0 * 64 * st["sum_nq_nt"] / bf_n                         # one 64-d dot product per (query, train)
        bytes_ = (st["sum_nq_plus_nt"] * 64 * 4) / bf_n
        extra["bf_l2_mfma"] = dict(kernel="k_bf_mfma_d64", bound="mfma", achieved=round(flops / dur / 1e12, 3), peak=FP32_PEAK_TFLOPS,
                                   unit="TFLOP/s", frac=round(flops / dur / 1e12 / FP32_PEAK_TFLOPS, 4), traffic=pmc_traffic("k_bf_mfma_d64")[0],
                                   avg_launch_ms=round(dur * 1e3, 4), flops_per_launch=flops, launches=bf_n,
                                   note="v_mfma_f32_32x32x2_f32 candidate filter (peak = dense f32 MFMA); the exact distances are "
                                        "evaluated by k_bf_verify_d64 for the few surviving candidates")
        extra["bf_l2_hbm"] = dict(bound="hbm", achieved=round(bytes_ / dur / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                                  frac=round(bytes_ / dur / 1e9 / HBM_PEAK_GBS, 5), bytes_per_launch=bytes_)
    in_ms, in_n = prof.get("integral", (0.0, 0))
    if in_n:
        # algorithmic bytes of cv::integral per ROI: h*w (u8 in) + 4 (h+1)(w+1) (i32 out) ~ 5 B/px
        px = st["roi_px"] / in_n
        b = px * 5.0
        extra["integral_hbm"] = dict(kernel="k_integral_rows+k_integral_cols", bound="hbm", achieved=round(b / (in_ms / in_n * 1e-3) / 1e9, 2),
                                     peak=HBM_PEAK_GBS, unit="GB/s", frac=round(b / (in_ms / in_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                     bytes_per_launch=b, avg_launch_ms=round(in_ms / in_n, 4))

    # ---- CPU baseline: the oracle (a port, cv2 is not installable) on a bounded sample, rank 0 at N = 1 only ------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0 and args.method == "surf":
        from oracle import oracle as O
        O.build()
        cores = os.cpu_count() or 1
        dirs = grid.true_directions()
        S = min(args.cpu_sample, P)
        t1 = time.perf_counter()
        for k in range(S):
            A, B = tiles[k], tiles[k + 1]
            ra = isa.roi_rect(A.shape, dirs[k], "first", 0.2); rb = isa.roi_rect(B.shape, dirs[k], "second", 0.2)
            ka, da = O.surf_detect_describe(np.ascontiguousarray(A[ra[0]:ra[0] + ra[2], ra[1]:ra[1] + ra[3]]), nthreads=cores)
            kb, db = O.surf_detect_describe(np.ascontiguousarray(B[rb[0]:rb[0] + rb[2], rb[1]:rb[1] + rb[3]]), nthreads=cores)
            pairs = O.bf_l2_ratio_matches(da, db, 0.75, nthreads=cores)
            O.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pairs, 3)
        dt = time.perf_counter() - t1
        cpu = dict(value=round(S / dt, 4), unit="image-pairs/s", cores=cores, kind="port",
                   sample="first %d pairs of the same grid, one ROI attempt each at the true direction (oracle SURF+BF-L2+mode, "
                          "OpenMP over %d threads), %.1f s" % (S, cores, dt))

    if rank == 0:
        out = {
            "metric": "image-pairs/sec (2048x2048 grayscale, SURF+BF)" if args.method == "surf" else "image-pairs/sec (%s)" % args.method,
            "value": round(P * args.steps / elapsed, 3),
            "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "synthetic %dx%d grid of %dx%d u8 tiles, serpentine path, %d pairs; %s; roiRatio 0.2, direction 1, directIncre 1"
                                   % (args.rows, args.cols, args.tile, args.tile, P,
                                      {"surf": "SURF(100,4,3,64-d)+BF-L2 knn2 ratio 0.75 + mode vote", "orb": "ORB(5000,1.2,8)+BF-Hamming 1-NN + mode vote",
                                       "phase": "FFT phase correlation of the ROI strips"}[args.method]),
                       "pairs": P, "parallelism": "pairs%d" % world, "speculation_window": args.window},
            "max_abs_offset_error_px": max_err, "pairs_failed": n_failed,
            "h2d_ms_rank0": round(t_up * 1e3, 2),
            "value_incl_h2d": round(P * args.steps / (elapsed + t_up * args.steps), 3),   # if every step also had to upload its tiles (never `value`)
            "attempts_per_step": st["attempts"] / max(args.steps, 1), "batches_per_step": st["batches"] / max(args.steps, 1),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "stages": stages,
        }
        out.update(extra)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
